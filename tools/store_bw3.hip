// store_bw3.hip -- does finer work granularity help the 360-column store pattern?  K lanes share a sample's 360 columns
// (thread k of a sample writes columns c = k, k + K, ...); K = 1 is today's pattern.  Also: block size 64 / 128 / 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

// sample-fastest thread order: thread t -> sample t % N ... no: keep 64 consecutive samples per wave, split columns by wave
template <int K, int BS>
__global__ __launch_bounds__(BS) void pk(double* __restrict__ out, size_t N, double v)
{
  const size_t t = (size_t)blockIdx.x * BS + threadIdx.x;
  const size_t wave = t >> 6;
  const int lane = threadIdx.x & 63;
  const size_t tile = wave / K;       // 64-sample tile
  const int k = (int)(wave % K);      // which share of the columns
  const size_t s = tile * 64 + lane;
  if (s >= N) return;
#pragma unroll 10
  for (int c = k; c < 360; c += K) out[(size_t)c * N + s] = v + c;
}
// contiguous share: thread k writes columns [k * 360/K, (k+1) * 360/K)
template <int K, int BS>
__global__ __launch_bounds__(BS) void pkc(double* __restrict__ out, size_t N, double v)
{
  const size_t t = (size_t)blockIdx.x * BS + threadIdx.x;
  const size_t wave = t >> 6;
  const int lane = threadIdx.x & 63;
  const size_t tile = wave / K;
  const int k = (int)(wave % K);
  const size_t s = tile * 64 + lane;
  if (s >= N) return;
  constexpr int W = 360 / K;
#pragma unroll 10
  for (int c = k * W; c < (k + 1) * W; ++c) out[(size_t)c * N + s] = v + c;
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
#define RUN(KERN, K, BS)                                                                                         \
  {                                                                                                              \
    const unsigned g = (unsigned)((N * K + BS - 1) / BS);                                                        \
    char nm[64];                                                                                                 \
    std::snprintf(nm, sizeof nm, #KERN " K=%d block=%d", K, BS);                                                 \
    report(nm, time_ms([&] { hipLaunchKernelGGL((KERN<K, BS>), dim3(g), dim3(BS), 0, nullptr, d, N, 1.0); }, 10)); \
  }
int main()
{
  const size_t N = 1000192;
  const size_t bytes = N * 360 * sizeof(double);
  double* d = nullptr;
  CHECK(hipMalloc((void**)&d, bytes));
  auto report = [&](const char* name, double ms) { std::printf("%-32s %8.3f ms  %6.3f TB/s\n", name, ms, bytes / ms * 1e-9); };
  for (int rep = 0; rep < 2; ++rep)
  {
    RUN(pk, 1, 256) RUN(pk, 1, 128) RUN(pk, 1, 64)
    RUN(pk, 2, 256) RUN(pk, 3, 256) RUN(pk, 4, 256) RUN(pk, 6, 256) RUN(pk, 12, 256)
    RUN(pkc, 2, 256) RUN(pkc, 3, 256) RUN(pkc, 4, 256) RUN(pkc, 6, 256) RUN(pkc, 12, 256)
    RUN(pk, 4, 64) RUN(pkc, 6, 64)
  }
  CHECK(hipFree(d));
  return 0;
}
