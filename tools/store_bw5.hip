// store_bw5.hip -- same 2.88 GB, same kernel shape (one thread per row, C stores of 512 B per wave), different number of
// concurrent column streams: C columns of N = 3.6e8 / C doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_columns(double* __restrict__ out, size_t N, int C, double v)
{
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
#pragma unroll 10
  for (int c = 0; c < C; ++c) out[(size_t)c * N + s] = v + c;
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
  const size_t total = 360000000ull;
  double* d = nullptr;
  CHECK(hipMalloc((void**)&d, total * sizeof(double)));
  for (int rep = 0; rep < 2; ++rep)
    for (int C : {1, 2, 5, 10, 20, 60, 120, 360, 720})
    {
      const size_t N = total / C;
      const unsigned g = (unsigned)((N + 255) / 256);
      const double ms = time_ms([&] { hipLaunchKernelGGL(k_columns, dim3(g), dim3(256), 0, nullptr, d, N, C, 1.0); }, 10);
      std::printf("%4d columns of %10zu doubles: %7.3f ms  %6.3f TB/s\n", C, N, ms, total * 8.0 / ms * 1e-9);
    }
  CHECK(hipFree(d));
  return 0;
}
