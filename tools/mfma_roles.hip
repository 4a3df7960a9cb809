// tools/mfma_roles.hip -- prices "two consumer waves per sweeper" (lever i) before anything is built: one workgroup
// per CU whose first four waves run an fp64 VALU stream (the sweeper's share of a 16-sample tile: VALU_PER_TILE v_fma_f64 in eight
// independent chains) and whose other waves run v_mfma_f64_16x16x4_f64 streams (the consumer's share: MFMA_PER_TILE per tile, split
// over the consumers of a SIMD).  Waves are dealt to the four SIMDs cyclically, so SIMD s hosts waves s, s + 4 (, s + 8).
// No synchronisation between the roles: the time per tile of the slower role is the floor of any kernel with that split.
//   A: 8 waves  = 1 VALU + 1 MFMA wave per SIMD (the shipped k_regressor_gram_duo)
//   B: 12 waves = 1 VALU + 2 MFMA waves per SIMD, each MFMA wave half of the tile's MFMAs
//   C: 8 waves, MFMA only / VALU only (each role alone on its SIMD: what it costs when the other is absent)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NWAVES, int VALU_PER_TILE, int MFMA_PER_TILE, bool RUN_VALU, bool RUN_MFMA>
__global__ __launch_bounds__(64 * NWAVES) void k(double* out, unsigned long long* cyc, int tiles, double seed)
{
  const int wave = threadIdx.x >> 6;
  const bool valu = wave < 4;
  constexpr int NCONS = (NWAVES - 4) / 4;  // MFMA waves per SIMD
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x;
  d4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = (d4){seed, seed, seed, seed};
  const double a = 1.0 + seed, b = seed * 0.5;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (valu)
  {
    if (RUN_VALU)
      for (int t = 0; t < tiles; ++t)
#pragma unroll 8
        for (int i = 0; i < VALU_PER_TILE / 8; ++i)
#pragma unroll
          for (int c = 0; c < 8; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
  }
  else if (RUN_MFMA)
  {
    for (int t = 0; t < tiles; ++t)
#pragma unroll 2
      for (int i = 0; i < MFMA_PER_TILE / NCONS / 10; ++i)
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 * NWAVES + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * NWAVES + wave] = c1 - c0;
}

template <int NWAVES, int V, int M, bool RV, bool RM>
static void run(const char* name, double* d_out, unsigned long long* d_cyc)
{
  const int blocks = 256, tiles = 200;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NWAVES, V, M, RV, RM>), dim3(blocks), dim3(64 * NWAVES), 0, nullptr, d_out, d_cyc, tiles, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL((k<NWAVES, V, M, RV, RM>), dim3(blocks), dim3(64 * NWAVES), 0, nullptr, d_out, d_cyc, tiles, 1e-3);
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * NWAVES);
  CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  std::vector<double> v, m;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < NWAVES; ++w) (w < 4 ? v : m).push_back((double)h[b * NWAVES + w] / tiles);
  std::sort(v.begin(), v.end());
  std::sort(m.begin(), m.end());
  std::printf("%-72s %7.3f ms   VALU wave %7.0f cyc/tile   MFMA wave %7.0f cyc/tile\n", name, ms, v[v.size() / 2], m.empty() ? 0.0 : m[m.size() / 2]);
}

int main()
{
  double* d_out;
  unsigned long long* d_cyc;
  CHECK(hipMalloc((void**)&d_out, sizeof(double) * 256 * 64 * 12));
  CHECK(hipMalloc((void**)&d_cyc, sizeof(unsigned long long) * 256 * 12));
  // 6 joints: 2 281 VALU instructions, 132 MFMAs per tile (rounded to multiples of 8 / 20); 7 joints: 2 650 / 192
  run<8, 2280, 140, true, true>("A  n=6: 1 VALU (2280 fma) + 1 MFMA wave (140) per SIMD", d_out, d_cyc);
  run<12, 2280, 140, true, true>("B  n=6: 1 VALU (2280 fma) + 2 MFMA waves (70 each) per SIMD", d_out, d_cyc);
  run<8, 2280, 140, true, false>("C  n=6: VALU wave alone", d_out, d_cyc);
  run<8, 2280, 140, false, true>("C  n=6: MFMA wave alone (140)", d_out, d_cyc);
  run<12, 2280, 140, false, true>("C  n=6: two MFMA waves alone (70 each)", d_out, d_cyc);
  run<8, 2648, 200, true, true>("A  n=7: 1 VALU (2648 fma) + 1 MFMA wave (200) per SIMD", d_out, d_cyc);
  run<12, 2648, 200, true, true>("B  n=7: 1 VALU (2648 fma) + 2 MFMA waves (100 each) per SIMD", d_out, d_cyc);
  // pass B of the robust factor at 7 joints: 384 MFMAs per tile
  run<8, 2648, 380, true, true>("A  pass B n=7: 1 VALU (2648 fma) + 1 MFMA wave (380) per SIMD", d_out, d_cyc);
  run<12, 2648, 380, true, true>("B  pass B n=7: 1 VALU (2648 fma) + 2 MFMA waves (190 each) per SIMD", d_out, d_cyc);
  return 0;
}
