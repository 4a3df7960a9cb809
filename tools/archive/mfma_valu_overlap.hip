// Do fp64 MFMA and fp64 VALU FMA overlap on gfx950?  One wave per SIMD (1024 waves), three loops: MFMA only, FMA only,
// both interleaved 1 MFMA : 16 FMA (equal pipe time if v_mfma_f64_16x16x4_f64 = 64 cycles and v_fma_f64 = 4 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 1 = MFMA, 2 = FMA, 3 = both
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed)
{
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, seed, seed, seed};
  double a = seed + threadIdx.x, b = seed * 0.5;
  double f[16];
  for (int i = 0; i < 16; ++i) f[i] = seed + i;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
      if (MODE & 1) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
      if (MODE & 2)
      {
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = fma(f[i], a, b);
      }
      if (MODE == 3)
      {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
double run(double* d, int blocks, int iters)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1e-3);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}
int main()
{
  double* d;
  const int iters = 20000;
  for (int wpe : {1, 2})
  {
    const int blocks = 1024 * wpe;
    CHECK(hipMalloc((void**)&d, sizeof(double) * 64 * blocks));
    const double m = run<1>(d, blocks, iters), f = run<2>(d, blocks, iters), both = run<3>(d, blocks, iters);
    const double mf = (double)blocks * iters * 4 * 2048 / (m * 1e-3) * 1e-12, ff = (double)blocks * iters * 64 * 64 * 2 / (f * 1e-3) * 1e-12;
    std::printf("%d wave(s)/SIMD: MFMA only %.3f ms (%.1f TF)   FMA only %.3f ms (%.1f TF)   interleaved %.3f ms   sum %.3f  max %.3f\n", wpe, m, mf,
                f, ff, both, m + f, m > f ? m : f);
    CHECK(hipFree(d));
  }
  return 0;
}
