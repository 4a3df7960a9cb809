// fp64 issue costs on gfx950 in SHADER CYCLES (s_memtime) with the in-kernel clock (s_memtime / s_memrealtime), one or two
// waves per SIMD: dependent / independent v_fma_f64 chains, v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 on 1..10
// accumulators, MFMA + FMA interleaved, ds_read_b128 + FMA mixes.  Answers: is an fp64 MFMA 64 cycles at a throttled clock or
// ~130 cycles at full clock; what a lone wave pays per dependent fp64 instruction; what LDS-fed FMA streams sustain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

// MODE 0: NCH independent dependent-chains of v_fma_f64 (NCH = 1: pure latency)
// MODE 1: NCH accumulators of v_mfma_f64_16x16x4_f64
// MODE 2: NCH accumulators of v_mfma_f64_4x4x4_4b_f64
// MODE 3: 1 MFMA (4 accumulators round robin) + NCH independent FMAs per group
// MODE 5: NCH v_mov_b32 (32-bit VALU) between two fp64 FMAs: does a lone wave pay 4 cycles for them
template <int MODE, int NCH>
__global__ __launch_bounds__(64) void k(double* out, Stamp* st, int iters, double seed)
{
  __shared__ double lds[64 * 2 * 8];
  for (int i = threadIdx.x; i < 64 * 2 * 8; i += 64) lds[i] = seed * i;
  __syncthreads();
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x;
  d4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = (d4){seed, seed, seed, seed};
  double a = 1.0 + seed, b = seed * 0.5;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
  {
    if (MODE == 0)
    {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    else if (MODE == 1)
    {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    else if (MODE == 2)
    {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) x[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x[c], 0, 0, 0);
    }
    else if (MODE == 3)
    {
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c & 15]) : "v"(a), "v"(b));
      }
    }
    else if (MODE == 5)
    {
#pragma unroll
      for (int u = 0; u < 8; ++u)
      {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[1]) : "v"(a), "v"(b));
#pragma unroll
        for (int c = 0; c < NCH; ++c) asm volatile("v_mov_b32 %0, %1" : "=v"(((int*)&x[2 + (c & 7)])[0]) : "v"(it));
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  double s = lds[threadIdx.x];
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0)
  {
    st[blockIdx.x].cyc = c1 - c0;
    st[blockIdx.x].real = t1 - t0;
  }
}

// LDS-fed FMA stream written in plain C++ so that the compiler pipelines it: per step R ds_read_b128 (2 doubles each) feed
// F FMAs into F accumulators; the lane's read addresses differ per lane (column offsets), rows advance by a constant.
template <int R, int F>
__global__ __launch_bounds__(64) void k_lds_fma(double* out, Stamp* st, int iters, double seed)
{
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int rows = 48, rowlen = 32;  // 48 rows of 32 doubles = 12 KB (8 workgroups per CU fit)
  for (int i = threadIdx.x; i < rows * rowlen; i += 64) tile[i] = seed * (i % 97);
  __syncthreads();
  double acc[F];
  for (int i = 0; i < F; ++i) acc[i] = 0.0;
  const int ca = (threadIdx.x & 7) * 2, cb = (threadIdx.x >> 3) * 2;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll 4
    for (int r = 0; r < rows; ++r)
    {
      const double* row = tile + r * rowlen;
      double v[2 * R];
#pragma unroll
      for (int i = 0; i < R; ++i)
      {
        const int col = (i & 1) ? cb + 16 * (i >> 1) : ca + 16 * (i >> 1);
        const double2 t = *(const double2*)(row + (col & 30));
        v[2 * i] = t.x;
        v[2 * i + 1] = t.y;
      }
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = fma(v[f % (2 * R)], v[(f * 7 + 1) % (2 * R)], acc[f]);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < F; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0)
  {
    st[blockIdx.x].cyc = c1 - c0;
    st[blockIdx.x].real = t1 - t0;
  }
}

static double* d_out;
static Stamp* d_st;

static void report(const char* name, int blocks, int iters, double ops_per_iter, float ms, double flop_per_op)
{
  std::vector<Stamp> h(blocks);
  CHECK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
  std::vector<double> cyc(blocks), clk(blocks);
  for (int i = 0; i < blocks; ++i)
  {
    cyc[i] = (double)h[i].cyc;
    clk[i] = (double)h[i].cyc / (double)h[i].real * 100.0;  // MHz (s_memrealtime = 100 MHz)
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double mc = cyc[blocks / 2], mk = clk[blocks / 2];
  std::printf("%-46s %8.3f ms  %7.1f cyc/op  clock %6.0f MHz  %6.1f TFLOP/s\n", name, ms, mc / (iters * ops_per_iter), mk,
              flop_per_op * ops_per_iter * iters * (double)blocks / (ms * 1e-3) * 1e-12);
}

template <int MODE, int NCH>
static void run(const char* name, int wpe, int iters, double ops_per_iter, double flop_per_op)
{
  const int blocks = 1024 * wpe;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, NCH>), dim3(blocks), dim3(64), 0, nullptr, d_out, d_st, iters, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL((k<MODE, NCH>), dim3(blocks), dim3(64), 0, nullptr, d_out, d_st, iters, 1e-3);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  char buf[128];
  std::snprintf(buf, sizeof buf, "%s, %d wave(s)/SIMD", name, wpe);
  report(buf, blocks, iters, ops_per_iter, ms, flop_per_op);
}

template <int R, int F>
static void run_lds(int wpe, int iters)
{
  const int blocks = 1024 * wpe;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  CHECK(hipFuncSetAttribute((const void*)k_lds_fma<R, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  const size_t lds = 48 * 32 * 8;
  hipLaunchKernelGGL((k_lds_fma<R, F>), dim3(blocks), dim3(64), lds, nullptr, d_out, d_st, iters, 1e-3);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL((k_lds_fma<R, F>), dim3(blocks), dim3(64), lds, nullptr, d_out, d_st, iters, 1e-3);
  CHECK(hipEventRecord(b, nullptr));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  char buf[128];
  std::snprintf(buf, sizeof buf, "LDS-fed FMA: %d ds_read_b128 + %d FMA per row, %d w/SIMD", R, F, wpe);
  report(buf, blocks, iters, 48.0 * (R + F), ms, 128.0 * F / (R + F));
}

int main()
{
  CHECK(hipMalloc((void**)&d_out, sizeof(double) * 64 * 4096));
  CHECK(hipMalloc((void**)&d_st, sizeof(Stamp) * 4096));
  for (int wpe : {1, 2})
  {
    if (wpe == 1)
    {
      run<0, 1>("v_fma_f64 1 chain (latency)", 1, 4000, 16, 128);
      run<0, 2>("v_fma_f64 2 chains", 1, 4000, 32, 128);
      run<0, 4>("v_fma_f64 4 chains", 1, 2000, 64, 128);
      run<0, 8>("v_fma_f64 8 chains", 1, 1000, 128, 128);
      run<1, 1>("mfma_f64_16x16x4 1 acc (latency)", 1, 2000, 4, 2048);
      run<1, 2>("mfma_f64_16x16x4 2 acc", 1, 1000, 8, 2048);
      run<1, 4>("mfma_f64_16x16x4 4 acc", 1, 1000, 16, 2048);
      run<1, 10>("mfma_f64_16x16x4 10 acc", 1, 500, 40, 2048);
      run<2, 1>("mfma_f64_4x4x4_4b 1 acc (latency)", 1, 2000, 4, 512);
      run<2, 4>("mfma_f64_4x4x4_4b 4 acc", 1, 1000, 16, 512);
      run<2, 8>("mfma_f64_4x4x4_4b 8 acc", 1, 1000, 32, 512);
      run<3, 4>("1 MFMA16 + 4 FMA (cyc per group)", 1, 1000, 4, 2048 + 4 * 128);
      run<3, 8>("1 MFMA16 + 8 FMA (cyc per group)", 1, 1000, 4, 2048 + 8 * 128);
      run<3, 16>("1 MFMA16 + 16 FMA (cyc per group)", 1, 1000, 4, 2048 + 16 * 128);
      run<3, 32>("1 MFMA16 + 32 FMA (cyc per group)", 1, 500, 4, 2048 + 32 * 128);
      run<5, 0>("2 FMA + 0 v_mov_b32 (cyc per group)", 1, 2000, 8, 256);
      run<5, 2>("2 FMA + 2 v_mov_b32 (cyc per group)", 1, 2000, 8, 256);
      run<5, 4>("2 FMA + 4 v_mov_b32 (cyc per group)", 1, 2000, 8, 256);
      run<5, 8>("2 FMA + 8 v_mov_b32 (cyc per group)", 1, 2000, 8, 256);
    }
    else
    {
      run<0, 1>("v_fma_f64 1 chain", 2, 4000, 16, 128);
      run<0, 4>("v_fma_f64 4 chains", 2, 2000, 64, 128);
      run<1, 4>("mfma_f64_16x16x4 4 acc", 2, 1000, 16, 2048);
      run<1, 10>("mfma_f64_16x16x4 10 acc", 2, 500, 40, 2048);
      run<3, 16>("1 MFMA16 + 16 FMA (cyc per group)", 2, 1000, 4, 2048 + 16 * 128);
      run<5, 4>("2 FMA + 4 v_mov_b32 (cyc per group)", 2, 2000, 8, 256);
      run<5, 8>("2 FMA + 8 v_mov_b32 (cyc per group)", 2, 2000, 8, 256);
    }
    run_lds<2, 8>(wpe, 40);
    run_lds<4, 16>(wpe, 40);
    run_lds<4, 32>(wpe, 20);
    run_lds<6, 40>(wpe, 20);
    run_lds<8, 40>(wpe, 20);
  }
  return 0;
}
