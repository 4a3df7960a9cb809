// store_bw4.hip -- 360-column store pattern with the natural block order vs an XCD-contiguous order (workgroup b runs on
// XCD b % 8: give every XCD one contiguous eighth of the samples) vs a rotated column order per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(double* __restrict__ out, size_t N, double v)
{
  size_t b = blockIdx.x;
  if (MODE == 1) b = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous (grid multiple of 8)
  const size_t s = b * 256 + threadIdx.x;
  if (s >= N) return;
  if (MODE == 2)
  {
    int c0 = (int)((blockIdx.x * 37u) % 360u);
    for (int i = 0; i < 360; ++i)
    {
      int c = c0 + i;
      if (c >= 360) c -= 360;
      out[(size_t)c * N + s] = v + c;
    }
  }
  else
  {
#pragma unroll 10
    for (int c = 0; c < 360; ++c) out[(size_t)c * N + s] = v + c;
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
  const size_t N = 3912 * 256;  // grid multiple of 8
  const size_t bytes = N * 360 * sizeof(double);
  double* d = nullptr;
  CHECK(hipMalloc((void**)&d, bytes));
  const unsigned g = (unsigned)(N / 256);
  auto report = [&](const char* name, double ms) { std::printf("%-40s %8.3f ms  %6.3f TB/s\n", name, ms, bytes / ms * 1e-9); };
  for (int rep = 0; rep < 3; ++rep)
  {
    report("natural block order", time_ms([&] { hipLaunchKernelGGL((k<0>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("XCD-contiguous sample ranges", time_ms([&] { hipLaunchKernelGGL((k<1>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("column order rotated per workgroup", time_ms([&] { hipLaunchKernelGGL((k<2>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
  }
  CHECK(hipFree(d));
  return 0;
}
