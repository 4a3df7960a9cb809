// tools/probe_scatter.cpp -- is the output-placement lottery (profiles/r2/placement.txt) about physical contiguity?  The stacked regressor
// launch (n = 6, N = 1e6) into (a) hipMalloc, (b) hipDeviceMallocContiguous, (c) a virtual range backed by hipMemCreate chunks of C bytes mapped
// in SHUFFLED order (physically scattered at chunk granularity), for several C.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probe_scatter.cpp -o tools/_build/probe_scatter \
//       -Lrosdyn_amd -lrdyn_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$PWD/rosdyn_amd
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "rdyn.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

static std::string read_file(const char* p)
{
  FILE* f = std::fopen(p, "rb");
  if (!f) { std::printf("cannot read %s\n", p); std::exit(1); }
  std::string s;
  char buf[4096];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
  std::fclose(f);
  return s;
}

// virtual range of `bytes` backed by chunks of `chunk` bytes created in order and mapped in shuffled order (shuffle = false: in order)
static void* vmm_alloc(size_t bytes, size_t chunk, bool shuffle, std::vector<hipMemGenericAllocationHandle_t>& handles)
{
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  void* base = nullptr;
  CHECK(hipMemAddressReserve(&base, n * chunk, 2 << 20, nullptr, 0));
  handles.resize(n);
  for (size_t i = 0; i < n; ++i) CHECK(hipMemCreate(&handles[i], chunk, &prop, 0));
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  if (shuffle)
  {
    std::mt19937_64 rng(12345);
    std::shuffle(order.begin(), order.end(), rng);
  }
  for (size_t i = 0; i < n; ++i) CHECK(hipMemMap((char*)base + i * chunk, chunk, 0, handles[order[i]], 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHECK(hipMemSetAccess(base, n * chunk, &acc, 1));
  return base;
}

int main()
{
  const int64_t N = 1000000;
  const int n = 6, P = 60;
  rdyn_chain* c = nullptr;
  const double g[3] = {0, 0, -9.806};
  if (rdyn_chain_from_urdf(read_file("tests/fixtures/ur10_like.urdf").c_str(), "base_link", "wrist_3_link", g, &c) != RDYN_OK) { std::printf("%s\n", rdyn_last_error()); return 1; }
  std::vector<double> h((size_t)3 * N * n);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U(-1, 1);
  for (auto& x : h) x = U(rng);
  double* d_in = nullptr;
  CHECK(hipMalloc((void**)&d_in, h.size() * 8));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  rdyn_batch b = {};
  b.n_samples = N; b.q = d_in; b.dq = d_in + N * n; b.ddq = d_in + 2 * N * n; b.layout = RDYN_LAYOUT_SAMPLE_MAJOR; b.device = -1;
  rdyn_regressor_layout yl = {n, 1, N * n};
  const size_t bytes = (size_t)P * N * n * 8;
  auto run = [&](void* Y) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (rdyn_regressor(c, &b, nullptr, (double*)Y, &yl) != RDYN_OK) { std::printf("%s\n", rdyn_last_error()); std::exit(1); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 6; ++i) rdyn_regressor(c, &b, nullptr, (double*)Y, &yl);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 6 * 1e3;
  };
  for (int rep = 0; rep < 2; ++rep)
  {
    std::printf("hipMalloc          :");
    for (int k = 0; k < 5; ++k) { void* p; CHECK(hipMalloc(&p, bytes)); std::printf(" %4.0f", run(p)); }   // kept alive
    std::printf("\ncontiguous         :");
    for (int k = 0; k < 3; ++k) { void* p; CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous)); std::printf(" %4.0f", run(p)); }
    std::printf("\n");
    for (size_t chunk : {(size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)512 << 20})
      for (int shuffle = 1; shuffle >= 0; --shuffle)
      {
        std::printf("vmm chunk %4zu MB %s:", chunk >> 20, shuffle ? "shuffled" : "in order");
        for (int k = 0; k < 3; ++k)
        {
          std::vector<hipMemGenericAllocationHandle_t> hs;
          void* p = vmm_alloc(bytes, chunk, shuffle != 0, hs);
          std::printf(" %4.0f", run(p));
        }
        std::printf("\n");
      }
  }
  return 0;
}
