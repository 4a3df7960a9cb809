// Peak rate of v_mfma_f64_16x16x4_f64 on gfx950 vs the number of independent accumulators per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){seed, seed, seed, seed};
  double a = seed + threadIdx.x, b = seed * 0.5;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
void run(int wpe)
{
  const int blocks = 1024 * wpe, iters = 40000 / NACC;
  double* d;
  CHECK(hipMalloc((void**)&d, sizeof(double) * 64 * blocks));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1e-3);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double n_mfma = (double)blocks * iters * NACC;
  std::printf("%2d accumulators, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s  (%.0f ns per MFMA per SIMD)\n", NACC, wpe, ms, n_mfma * 2048 / (ms * 1e-3) * 1e-12,
              ms * 1e6 / (n_mfma / 1024));
  CHECK(hipFree(d));
}
int main()
{
  for (int wpe : {1, 2, 4})
  {
    run<1>(wpe); run<2>(wpe); run<4>(wpe); run<8>(wpe); run<16>(wpe);
  }
  return 0;
}
