// store_bw.hip -- pure store streams on MI355X: the write ceiling of the regressor's output patterns.  Five probes in one binary
// (results: profiles/r1/store_bw.txt, profiles/r2/image_ab.txt):
//   store_bw 1   store width, nontemporal or not, contiguous vs 360 concurrent column streams, grid size   (default)
//   store_bw 2   candidate output orders of the (N x 360) regressor: per-sample image, wave-tiled, ...
//   store_bw 3   K lanes share a sample's 360 columns; block sizes 64 / 128 / 256
//   store_bw 4   natural block order vs XCD-contiguous sample ranges vs rotated column order
//   store_bw 5   the same 2.88 GB with C concurrent column streams
// build: hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o tools/_build/store_bw
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <vector>

// store_bw.hip -- what write bandwidth does this MI355X give a pure store stream?  (ceiling for the regressor kernel,
// whose traffic is 94 % writes).  Variants: store width, nontemporal or not, contiguous vs many concurrent column streams
// (the element-major regressor writes 360 columns of N doubles at once), grid size.
namespace probe1
{
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

int run(int argc, char** argv)
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
}  // namespace probe1

// store_bw2.hip -- candidate output orders for the regressor (N samples x 360 doubles), pure store streams.
//  P1 per-sample image: sample s owns 2880 contiguous bytes; a wave (64 samples) writes, link by link (6 links), the
//     64 segments of 480 B of that link as 60 instructions over the flat (sample, k) list (what an LDS transpose gives)
//  P2 wave-tiled: [wave tile of 64 samples][360][64]: every instruction 512 contiguous bytes, a wave's 184 KB in order
//  P3 stacked column-major (6N x 60): per column a wave's 64 samples are 3 KB contiguous = 6 instructions
//  P0 element-major columns (today's default): 360 columns, 512 B per instruction
namespace probe2
{
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

int run(int argc, char** argv)
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
}  // namespace probe2

// store_bw3.hip -- does finer work granularity help the 360-column store pattern?  K lanes share a sample's 360 columns
// (thread k of a sample writes columns c = k, k + K, ...); K = 1 is today's pattern.  Also: block size 64 / 128 / 256.
namespace probe3
{
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
int run(int argc, char** argv)
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
}  // namespace probe3

// store_bw4.hip -- 360-column store pattern with the natural block order vs an XCD-contiguous order (workgroup b runs on
// XCD b % 8: give every XCD one contiguous eighth of the samples) vs a rotated column order per workgroup.
namespace probe4
{
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
int run(int argc, char** argv)
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
}  // namespace probe4

// store_bw5.hip -- same 2.88 GB, same kernel shape (one thread per row, C stores of 512 B per wave), different number of
// concurrent column streams: C columns of N = 3.6e8 / C doubles.
namespace probe5
{
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
int run(int argc, char** argv)
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
}  // namespace probe5


// probe 6 (round 3) -- the EXACT copy-out instruction shapes of k_image_sweep<6, 0>, stores only (no sweep, no input reads), into K
// output allocations kept alive: what the store schedule alone reaches in the fast and in the slow placement state.
//   stacked     wave w owns samples 64 w ..: per link 10 columns x 3 (64 lanes x 16 B = 1 KiB) nontemporal stores at column offset 3072 w
//   per-sample  wave w owns 64 images of 2 880 B (one contiguous 184 320-byte region): per link, for every image the whole 128-byte
//               lines completed by that link, 2 images x 32 lanes x 16 B per instruction
namespace probe6
{
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(64) void k_stacked(double* __restrict__ out, size_t N)
{
  const size_t w = blockIdx.x;
  const int lane = threadIdx.x;
  const d2u v = {1.0, 2.0};
  for (int f = 0; f < 6; ++f)
    for (int pp = 0; pp < 10; ++pp)
    {
      char* col = (char*)(out + (size_t)(10 * f + pp) * (N * 6) + w * 64 * 6);
#pragma unroll
      for (int it = 0; it < 3; ++it) __builtin_nontemporal_store(v, (d2u*)(col + (it * 64 + lane) * 16));
    }
}
// the same 3 KB per column and wave with 8 B per lane (six 512-byte stores) -- "dwordx4 stores are slower than dwordx2" for a plain fill
__global__ __launch_bounds__(64) void k_stacked8(double* __restrict__ out, size_t N)
{
  const size_t w = blockIdx.x;
  const int lane = threadIdx.x;
  for (int f = 0; f < 6; ++f)
    for (int pp = 0; pp < 10; ++pp)
    {
      double* col = out + (size_t)(10 * f + pp) * (N * 6) + w * 64 * 6;
#pragma unroll
      for (int it = 0; it < 6; ++it) __builtin_nontemporal_store(1.0, col + it * 64 + lane);
    }
}
__global__ __launch_bounds__(64) void k_images(double* __restrict__ out, size_t N)
{
  const size_t w = blockIdx.x;
  const int lane = threadIdx.x, sub = lane >> 5, ch = lane & 31;
  char* const base = (char*)out + w * (64 * 2880);
  const d2u v = {1.0, 2.0};
  for (int f = 0; f < 6; ++f)
    for (int it = 0; it < 32; ++it)
    {
      const unsigned i = 2 * it + sub;
      const unsigned lo = (i * 2880u + 480u * f) & ~127u;
      const unsigned hi = (f == 5 && i == 63) ? 64u * 2880u : ((i * 2880u + 480u * (f + 1)) & ~127u);
      const unsigned x = lo + 16u * ch;
      if (x < hi) __builtin_nontemporal_store(v, (d2u*)(base + x));
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
int run(int argc, char** argv)
{
  const size_t N = 1000000 - 1000000 % 64;  // whole waves
  const int K = argc > 1 ? std::atoi(argv[1]) : 8;
  std::vector<double*> bufs;
  for (int i = 0; i < K; ++i)
  {
    double* d = nullptr;
    CHECK(hipMalloc((void**)&d, (size_t)1000000 * 360 * sizeof(double)));
    bufs.push_back(d);
  }
  const unsigned g = (unsigned)(N / 64);
  const double bytes = (double)N * 2880.0;
  for (int i = 0; i < K; ++i)
  {
    double* d = bufs[(size_t)i];
    const double ms_s = time_ms([&] { hipLaunchKernelGGL(k_stacked, dim3(g), dim3(64), 0, nullptr, d, N); }, 10);
    const double ms_i = time_ms([&] { hipLaunchKernelGGL(k_images, dim3(g), dim3(64), 0, nullptr, d, N); }, 10);
    const double ms_8 = time_ms([&] { hipLaunchKernelGGL(k_stacked8, dim3(g), dim3(64), 0, nullptr, d, N); }, 10);
    const double ms_f = time_ms([&] { CHECK(hipMemsetAsync(d, 0, (size_t)bytes, nullptr)); }, 10);
    std::printf("buffer %d: stacked pattern %7.1f us %6.3f TB/s (8 B per lane: %7.1f us) | per-sample pattern %7.1f us %6.3f TB/s | hipMemset %7.1f us %6.3f TB/s\n", i,
                ms_s * 1e3, bytes / ms_s * 1e-9, ms_8 * 1e3, ms_i * 1e3, bytes / ms_i * 1e-9, ms_f * 1e3, bytes / ms_f * 1e-9);
  }
  for (double* d : bufs) CHECK(hipFree(d));
  return 0;
}
}  // namespace probe6

int main(int argc, char** argv)
{
  const int which = argc > 1 ? std::atoi(argv[1]) : 1;
  if (argc > 1) { --argc; ++argv; }  // the probe sees its own arguments from argv[1] on
  switch (which)
  {
  case 1: return probe1::run(argc, argv);
  case 2: return probe2::run(argc, argv);
  case 3: return probe3::run(argc, argv);
  case 4: return probe4::run(argc, argv);
  case 5: return probe5::run(argc, argv);
  case 6: return probe6::run(argc, argv);
  default: std::fprintf(stderr, "usage: store_bw [1-6]\n"); return 2;
  }
}
