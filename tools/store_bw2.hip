// store_bw2.hip -- candidate output orders for the regressor (N samples x 360 doubles), pure store streams.
//  P1 per-sample image: sample s owns 2880 contiguous bytes; a wave (64 samples) writes, link by link (6 links), the
//     64 segments of 480 B of that link as 60 instructions over the flat (sample, k) list (what an LDS transpose gives)
//  P2 wave-tiled: [wave tile of 64 samples][360][64]: every instruction 512 contiguous bytes, a wave's 184 KB in order
//  P3 stacked column-major (6N x 60): per column a wave's 64 samples are 3 KB contiguous = 6 instructions
//  P0 element-major columns (today's default): 360 columns, 512 B per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st(double* p, double v)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void p0(double* __restrict__ out, size_t N, double v)
{
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= N) return;
#pragma unroll 10
  for (int c = 0; c < 360; ++c) st<NT>(out + (size_t)c * N + s, v + c);
}
template <bool NT>
__global__ __launch_bounds__(256) void p1(double* __restrict__ out, size_t N, double v)
{
  const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;  // wave
  const int lane = threadIdx.x & 63;
  const size_t s0 = w * 64;
  if (s0 >= N) return;
  for (int f = 0; f < 6; ++f)
#pragma unroll 10
    for (int i = 0; i < 60; ++i)
    {
      const int flat = i * 64 + lane;         // 0 .. 3839 over (sample, k), k < 60
      const int smp = flat / 60, k = flat - smp * 60;
      st<NT>(out + (s0 + smp) * 360 + f * 60 + k, v + i);
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void p2(double* __restrict__ out, size_t N, double v)
{
  const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w * 64 >= N) return;
  double* o = out + w * 64 * 360 + lane;
#pragma unroll 10
  for (int c = 0; c < 360; ++c) st<NT>(o + c * 64, v + c);
}
template <bool NT>
__global__ __launch_bounds__(256) void p3(double* __restrict__ out, size_t N, double v)
{
  const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w * 64 >= N) return;
  for (int c = 0; c < 60; ++c)
  {
    double* o = out + (size_t)c * 6 * N + w * 384 + lane;
#pragma unroll
    for (int i = 0; i < 6; ++i) st<NT>(o + i * 64, v + c);
  }
}
// P4: block-tiled [block of 256 samples][360][256]: instruction = 512 B, block region 737 KB
template <bool NT>
__global__ __launch_bounds__(256) void p4(double* __restrict__ out, size_t N, double v)
{
  if ((size_t)blockIdx.x * 256 >= N) return;
  double* o = out + (size_t)blockIdx.x * 256 * 360 + threadIdx.x;
#pragma unroll 10
  for (int c = 0; c < 360; ++c) st<NT>(o + c * 256, v + c);
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
  const size_t N = 1000000 / 256 * 256 + 256;  // whole blocks
  const size_t bytes = N * 360 * sizeof(double);
  double* d = nullptr;
  CHECK(hipMalloc((void**)&d, bytes));
  const unsigned g = (unsigned)(N / 256);
  auto report = [&](const char* name, double ms) { std::printf("%-40s %8.3f ms  %6.3f TB/s\n", name, ms, bytes / ms * 1e-9); };
  for (int rep = 0; rep < 2; ++rep)
  {
    report("P0 element-major columns", time_ms([&] { hipLaunchKernelGGL((p0<false>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P0 element-major columns nt", time_ms([&] { hipLaunchKernelGGL((p0<true>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P1 per-sample image via flat list", time_ms([&] { hipLaunchKernelGGL((p1<false>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P1 per-sample image via flat list nt", time_ms([&] { hipLaunchKernelGGL((p1<true>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P2 wave-tiled [64]", time_ms([&] { hipLaunchKernelGGL((p2<false>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P2 wave-tiled [64] nt", time_ms([&] { hipLaunchKernelGGL((p2<true>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P3 stacked, 3 KB runs", time_ms([&] { hipLaunchKernelGGL((p3<false>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P3 stacked, 3 KB runs nt", time_ms([&] { hipLaunchKernelGGL((p3<true>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P4 block-tiled [256]", time_ms([&] { hipLaunchKernelGGL((p4<false>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
    report("P4 block-tiled [256] nt", time_ms([&] { hipLaunchKernelGGL((p4<true>), dim3(g), dim3(256), 0, nullptr, d, N, 1.0); }, 10));
  }
  CHECK(hipFree(d));
  return 0;
}
