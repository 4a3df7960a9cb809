// store_bw.hip -- what write bandwidth does this MI355X give a pure store stream?  (ceiling for the regressor kernel,
// whose traffic is 94 % writes).  Variants: store width, nontemporal or not, contiguous vs many concurrent column streams
// (the element-major regressor writes 360 columns of N doubles at once), grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int W, bool NT>
__global__ __launch_bounds__(256) void k_contig(double* __restrict__ out, size_t n_vec, double v)
{
  // grid-stride over W-double vectors
  typedef double vecT __attribute__((ext_vector_type(W)));
  vecT val;
  for (int i = 0; i < W; ++i) val[i] = v + i;
  vecT* o = reinterpret_cast<vecT*>(out);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256)
  {
    if (NT) __builtin_nontemporal_store(val, o + i);
    else o[i] = val;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_contig1(double* __restrict__ out, size_t n, double v)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}
// one thread per sample, C columns of N doubles (column c at out + c * N): every store instruction of a wave writes
// 512 contiguous bytes of one column -- the element-major regressor pattern
template <bool NT>
__global__ __launch_bounds__(256) void k_columns(double* __restrict__ out, size_t N, int C, double v)
{
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
#pragma unroll 8
  for (int c = 0; c < C; ++c)
  {
    if (NT) __builtin_nontemporal_store(v + c, out + (size_t)c * N + s);
    else out[(size_t)c * N + s] = v + c;
  }
}
// the same, but each thread owns two adjacent samples and stores 16 bytes (half as many threads)
template <bool NT>
__global__ __launch_bounds__(256) void k_columns2(double* __restrict__ out, size_t N, int C, double v)
{
  typedef double vec2 __attribute__((ext_vector_type(2)));
  const size_t s = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (s >= N) return;
  vec2 val = {v, v + 1};
#pragma unroll 8
  for (int c = 0; c < C; ++c)
  {
    vec2* p = reinterpret_cast<vec2*>(out + (size_t)c * N + s);
    if (NT) __builtin_nontemporal_store(val, p);
    else *p = val;
  }
}

template <class F>
static double time_ms(F&& launch, int reps)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main()
{
  const size_t N = 1000000, C = 360;           // 2.88 GB, the config-2 regressor
  const size_t n = N * C, bytes = n * sizeof(double);
  double* d = nullptr;
  CHECK(hipMalloc((void**)&d, bytes));
  const int reps = 10;
  auto report = [&](const char* name, double ms) { std::printf("%-44s %8.3f ms  %6.3f TB/s\n", name, ms, bytes / ms * 1e-9); };
  report("hipMemsetAsync", time_ms([&] { CHECK(hipMemsetAsync(d, 0, bytes, nullptr)); }, reps));
  for (int grid : {1024, 2048, 4096, 16384, 65536})
  {
    char nm[96];
    std::snprintf(nm, sizeof nm, "contig 8B   grid %6d", grid);
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_contig1<false>), dim3(grid), dim3(256), 0, nullptr, d, n, 1.0); }, reps));
    std::snprintf(nm, sizeof nm, "contig 8B nt grid %6d", grid);
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_contig1<true>), dim3(grid), dim3(256), 0, nullptr, d, n, 1.0); }, reps));
    std::snprintf(nm, sizeof nm, "contig 16B  grid %6d", grid);
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_contig<2, false>), dim3(grid), dim3(256), 0, nullptr, d, n / 2, 1.0); }, reps));
    std::snprintf(nm, sizeof nm, "contig 16B nt grid %6d", grid);
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_contig<2, true>), dim3(grid), dim3(256), 0, nullptr, d, n / 2, 1.0); }, reps));
    std::snprintf(nm, sizeof nm, "contig 32B nt grid %6d", grid);
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_contig<4, true>), dim3(grid), dim3(256), 0, nullptr, d, n / 4, 1.0); }, reps));
  }
  const unsigned g1 = (unsigned)((N + 255) / 256), g2 = (unsigned)((N / 2 + 255) / 256);
  report("360 columns, 8B/lane", time_ms([&] { hipLaunchKernelGGL((k_columns<false>), dim3(g1), dim3(256), 0, nullptr, d, N, (int)C, 1.0); }, reps));
  report("360 columns, 8B/lane nt", time_ms([&] { hipLaunchKernelGGL((k_columns<true>), dim3(g1), dim3(256), 0, nullptr, d, N, (int)C, 1.0); }, reps));
  report("360 columns, 16B/lane", time_ms([&] { hipLaunchKernelGGL((k_columns2<false>), dim3(g2), dim3(256), 0, nullptr, d, N, (int)C, 1.0); }, reps));
  report("360 columns, 16B/lane nt", time_ms([&] { hipLaunchKernelGGL((k_columns2<true>), dim3(g2), dim3(256), 0, nullptr, d, N, (int)C, 1.0); }, reps));
  CHECK(hipFree(d));
  return 0;
}
