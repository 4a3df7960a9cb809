// rdyn_tsqr_wide.hip -- Householder R factor for the shapes the register-resident folds of rdyn_tsqr.hip cannot hold
// (no reference counterpart: the identification step lives outside rosdyn_core, /root/reference/README.md:15).
//
// rdyn_tsqr.hip keeps the running factor AND the row block in the registers of one wave: at most 64 columns for a device matrix,
// 7 chain joints (71 columns) for the fused sweep, 6 joints once friction / spring columns ride along.  Wider problems --
//   * [Y | C | tau_meas] of a 7-joint arm (friction_polynomial1.h:126, ideal_spring.h:64 columns beside getRegressor): 86 columns,
//   * a caller's materialised matrix of up to 111 columns + right-hand side (what rdyn_gram already takes)
// -- run here, with the factor in LDS: one workgroup keeps its running R PACKED in LDS (column j holds its j + 1 entries, 51 KB at
// 112 columns) beside ONE row block and folds block after block into it, R <- qr([R ; block]), one barrier per column step, four
// threads per column.  A block is
//   k_regressor_tsqr_wide   the 16-sample tile the workgroup's first wave has just swept (the row-pair sweeper of rdyn_duo_gram.hip /
//                           rdyn_tsqr.hip writing a RECTANGULAR tile: every column 16 n rows, the structural zeros stored), or
//   k_tsqr_wide_rows        up to 128 rows of a column-major device matrix -- also the tree: the per-workgroup factors are folded
//                           sixteen at a time as the rows of a stacked matrix, in a fixed order (bitwise reproducible).
// A step is a dependent chain of ~900 cycles whatever the block holds (LDS round trips, one square root, one division): 86 steps per
// 16-sample tile = 35 us, 40 ms for 4e6 samples of a 7-joint arm with 14 component columns.  This is the slow, unconditionally
// robust route: small batches, and the STAND-BY of the preconditioned route of rdyn_cholqr.hip for the shapes rdyn_tsqr.hip does not
// serve (the device starts it only when that route cannot vouch for its result).  fp64 VALU + LDS only.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_duo_common.h"

#define DUO_BARRIER()  // the link body shared with rdyn_duo_gram.hip synchronises with its consumer wave there; not here

namespace
{
constexpr int NTW = 512;  // threads per workgroup: two waves per SIMD leave the sweeping wave its 256 registers

__device__ __forceinline__ int tri_off(int j) { return j * (j + 1) / 2; }

// R <- qr([R ; B]): R packed in LDS (Rp[tri_off(j) + i], i <= j), B = nrows x n1 column-major in LDS (leading dimension ldb doubles,
// ldb = 4 mod 8: the four lanes of a column and the sixteen columns of a wave read disjoint banks).  B is destroyed.
// Ends with a barrier.  Columns that are exactly zero (or rounding residue below 1e-140) in the block are passed.
__device__ __forceinline__ void wide_fold(double* Rp, double* B, int ldb, int nrows, int n1, int tid)
{
  const int lane = tid & 63;
  for (int k = 0; k < n1; ++k)
  {
    const double* const bk = B + (size_t)k * ldb;
    double sigma = 0.0;
    for (int r = lane; r < nrows; r += 64) sigma = fma(bk[r], bk[r], sigma);
    for (int o = 32; o > 0; o >>= 1) sigma += __shfl_xor(sigma, o);  // every wave: the same sum in the same order
    const double alpha = Rp[tri_off(k) + k];
    double beta = alpha;
    if (sigma > 1e-280)
    {
      const double norm = sqrt(fma(alpha, alpha, sigma));
      beta = alpha > 0.0 ? -norm : norm;
      const double v0 = alpha - beta, scale = 2.0 / fma(v0, v0, sigma);
      const int ncol = n1 - k - 1;
      for (int e = tid; e < (ncol * 4 + NTW - 1) / NTW * NTW; e += NTW)
      {
        const bool on = e < ncol * 4;
        const int j = on ? k + 1 + (e >> 2) : k, q = e & 3;
        double* const bj = B + (size_t)j * ldb;
        double* const rj = Rp + tri_off(j);
        double d = (on && q == 0) ? v0 * rj[k] : 0.0;
        if (on)
          for (int r = q; r < nrows; r += 4) d = fma(bk[r], bj[r], d);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        const double f = scale * d;
        if (on)
        {
          if (q == 0) rj[k] = fma(-f, v0, rj[k]);
          for (int r = q; r < nrows; r += 4) bj[r] = fma(-f, bk[r], bj[r]);
        }
      }
    }
    __syncthreads();
    if (tid == 0) Rp[tri_off(k) + k] = beta;
  }
  __syncthreads();
}

__device__ __forceinline__ void store_packed_factor(const double* Rp, double* out, int n1, int tid)
{
  for (int e = tid; e < n1 * n1; e += NTW)
  {
    const int i = e % n1, j = e / n1;
    out[e] = i <= j ? Rp[tri_off(j) + i] : 0.0;
  }
}

// ---------------------------------------------------------------- leaf: regressor rows from the workgroup's own sweep
// fa: the RECTANGULAR tile layout of rdyn_api.cpp (build_rect_tile): every column 16 n_active rows + 4 doubles of padding; link f at
// lds_off[f], component column k at lds_off_c + k * comp_stride + (its joint) * comp_row_step, tau_meas at lds_off_b.
// n1 = 10 n_joints + n_comp_cols + 1 columns.  factors: [gridDim.x][n1 * n1].
__global__ __launch_bounds__(NTW) void k_regressor_tsqr_wide(const RdynLdsGramArgs fa, int n_joints, int n1, double* __restrict__ factors)
{
  constexpr bool DIRECT = false, ALLREV = false;
  if (fa.run_flag && *fa.run_flag == 0) return;  // stand-by call, not needed (uniform: every wave leaves)
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  ChainPtr c = as_const(fa.chain);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tri = tri_off(n1);
  double* const Rp = (double*)lds_raw;
  char* const tile = lds_raw + (((size_t)tri * 8 + 255) & ~(size_t)255);
  const int n = fa.n_active, nrows = 16 * n, ldb = fa.lds_stride[0] / 8;
  for (int i = tid; i < tri; i += NTW) Rp[i] = 0.0;
  const int s_loc = lane >> 2, k = lane & 3;
  const int r0 = k, r1 = k + 4;
  int fB = n_joints;
  for (int f = n_joints - 1; f >= 0; --f)
    if (fa.lds_m[f] >= 5) fB = f;
  const int64_t t_mul = fa.tile_stride > 1 ? fa.tile_stride : 1;
  const int64_t n_tiles = ((fa.n_samples + 15) / 16 + t_mul - 1) / t_mul;
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x)
  {
    // the fold fills the structural zeros of the previous tile: every tile starts from a zero block
    for (int i = tid; i < n1 * ldb; i += NTW) ((double*)tile)[i] = 0.0;
    __syncthreads();
    if (wave == 0)
    {
      int64_t sx = tl * t_mul * 16 + s_loc;
      const bool valid = sx < fa.n_samples;
      if (!valid) sx = fa.n_samples - 1;
      const int64_t o = sx * fa.in_ss;
      double qa = 0.0, dqa = 0.0, ddqa = 0.0, qb = 0.0, dqb = 0.0, ddqb = 0.0, tb0 = 0.0, tb1 = 0.0;
      if (fa.bcol)
      {
        if (r0 < n) tb0 = fa.bcol[o + r0 * fa.in_sj];
        if (r1 < n) tb1 = fa.bcol[o + r1 * fa.in_sj];
      }
      if (k < n)
      {
        qa = fa.q[o + k * fa.in_sj];
        dqa = fa.dq[o + k * fa.in_sj];
        ddqa = fa.ddq[o + k * fa.in_sj];
      }
      if (k + 4 < n)
      {
        qb = fa.q[o + (k + 4) * fa.in_sj];
        dqb = fa.dq[o + (k + 4) * fa.in_sj];
        ddqb = fa.ddq[o + (k + 4) * fa.in_sj];
      }
      if (!valid) tb0 = tb1 = 0.0;
      const int m0idx = valid ? r0 : -2, m1idx = valid ? r1 : -2;
      double sna, csa, snb, csb;
      sincos(qa, &sna, &csa);
      sincos(qb, &snb, &csb);
      const double oca = 1.0 - csa, ocb = 1.0 - csb;
      V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
      V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
      V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);
#pragma unroll 1
      for (int f = 0; f < n_joints; ++f)
      {
#include "rdyn_duo_link_body.inc"
      }
      if (fa.n_comp_cols > 0)
      {
#include "rdyn_duo_comp_cols.inc"
      }
      {
        char* const lb = tile + fa.lds_off_b + s_loc * 8;
        if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
        if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
      }
    }
    __syncthreads();
    wide_fold(Rp, (double*)tile, ldb, nrows, n1, tid);
  }
  __syncthreads();
  store_packed_factor(Rp, factors + (int64_t)blockIdx.x * ((int64_t)n1 * n1), n1, tid);
}

// ---------------------------------------------------------------- leaf / tree: row blocks of a column-major device matrix
// Rows of [A | b] (b may be null; n1 = n_cols + (b != null)).  seg_rows > 0: the rows come in segments of seg_rows rows, segment g at
// A + g * seg_stride (a slab of n1 x n1 factors with lda = n1: the tree levels); 0: one matrix.  Workgroup w folds the rows
// [w * rows_per_wg, (w + 1) * rows_per_wg) in blocks of rb rows and writes its factor to out + w * n1 * n1; workgroup 0 also folds
// `extra` (an n1 x n1 upper-triangular factor, column-major: the caller's running factor when accumulating), last.
__global__ __launch_bounds__(NTW) void k_tsqr_wide_rows(const double* __restrict__ A, const double* __restrict__ b, int64_t rows, int64_t lda, int n_cols,
                                                        int seg_rows, int64_t seg_stride, int rb, int64_t rows_per_wg, double* __restrict__ out,
                                                        const double* __restrict__ extra, const int* __restrict__ run_flag)
{
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int tid = threadIdx.x;
  const int n1 = n_cols + (b ? 1 : 0), tri = tri_off(n1), ldb = rb + 4;
  double* const Rp = (double*)lds_raw;
  double* const B = (double*)(lds_raw + (((size_t)tri * 8 + 255) & ~(size_t)255));
  for (int i = tid; i < tri; i += NTW) Rp[i] = 0.0;
  const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t row_hi = row_lo + rows_per_wg < rows ? row_lo + rows_per_wg : rows;
  __syncthreads();
  for (int64_t r0 = row_lo; r0 < row_hi; r0 += rb)
  {
    const int cnt = (int)(row_hi - r0 < rb ? row_hi - r0 : rb);
    for (int e = tid; e < n1 * rb; e += NTW)
    {
      const int col = e / rb, r = e - col * rb;
      double v = 0.0;
      if (r < cnt)
      {
        const int64_t gr = r0 + r;
        if (col >= n_cols)
          v = b[gr];
        else if (seg_rows > 0)
        {
          const int64_t g = gr / seg_rows;
          v = A[g * seg_stride + (int64_t)col * lda + (gr - g * seg_rows)];
        }
        else
          v = A[(int64_t)col * lda + gr];
      }
      B[(size_t)col * ldb + r] = v;
    }
    __syncthreads();
    wide_fold(Rp, B, ldb, rb, n1, tid);
  }
  if (extra && blockIdx.x == 0)
  {
    for (int r0 = 0; r0 < n1; r0 += rb)
    {
      for (int e = tid; e < n1 * rb; e += NTW)
      {
        const int col = e / rb, r = e - col * rb, gr = r0 + r;
        B[(size_t)col * ldb + r] = (gr < n1 && gr <= col) ? extra[(int64_t)col * n1 + gr] : 0.0;
      }
      __syncthreads();
      wide_fold(Rp, B, ldb, rb, n1, tid);
    }
  }
  __syncthreads();
  store_packed_factor(Rp, out + (int64_t)blockIdx.x * ((int64_t)n1 * n1), n1, tid);
}

hipError_t opt_in(const void* kernel, std::atomic<uint64_t>& done)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  return hipSuccess;
}

constexpr size_t kWideLdsBudget = 158 * 1024;
size_t tri_bytes(int n1) { return (((size_t)n1 * (n1 + 1) / 2) * 8 + 255) & ~(size_t)255; }

// rows per block of k_tsqr_wide_rows: what fits beside the packed factor, a multiple of 4, at most 128 (0: does not fit)
int wide_rows_per_block(int n1)
{
  const size_t left = kWideLdsBudget - tri_bytes(n1);
  int rb = (int)(left / ((size_t)n1 * 8)) - 4;
  rb &= ~3;
  if (rb > 128) rb = 128;
  return rb >= 16 ? rb : 0;
}

// count factors (n1 x n1, column-major, contiguous) at `in` -> R, sixteen per workgroup and level; scratch: room for 16 factors
hipError_t wide_tree(const double* in, int count, double* scratch, double* R, int n1, const double* extra, const int* run_flag, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in((const void*)k_tsqr_wide_rows, attr);
  if (e != hipSuccess) return e;
  const int rb = wide_rows_per_block(n1);
  if (rb == 0) return hipErrorInvalidValue;
  const size_t lds = tri_bytes(n1) + (size_t)n1 * (rb + 4) * 8;
  const int fan = 16;
  while (true)
  {
    const int nout = (count + fan - 1) / fan;
    const bool last = nout <= 1;
    hipLaunchKernelGGL(k_tsqr_wide_rows, dim3(last ? 1 : nout), dim3(NTW), lds, st, in, (const double*)nullptr, (int64_t)count * n1, (int64_t)n1, n1, n1,
                       (int64_t)n1 * n1, rb, (int64_t)fan * n1, last ? R : scratch, last ? extra : (const double*)nullptr, run_flag);
    e = hipGetLastError();
    if (e != hipSuccess || last) return e;
    in = scratch;
    count = nout;
    scratch = scratch + (size_t)nout * n1 * n1;  // (256 leaves -> 16 -> 1: the second level reads what the first wrote, writes R)
  }
}
}  // namespace

// widest factor the LDS-resident folds serve (right-hand side included)
int rdyn_tsqr_wide_max_cols() { return 112; }

// LDS of a k_regressor_tsqr_wide launch (0: the tile does not fit beside the factor)
size_t rdyn_regressor_tsqr_wide_lds_bytes(int n1, int n_active)
{
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols() || n_active < 1 || n_active > 8) return 0;
  const size_t bytes = tri_bytes(n1) + (size_t)n1 * (16 * n_active + 4) * 8;
  return bytes <= kWideLdsBudget ? bytes : 0;
}

// doubles of workspace: the leaves' factors + the tree's intermediate level
size_t rdyn_tsqr_wide_workspace_doubles(int n1, int blocks) { return (size_t)(blocks + 17) * n1 * n1; }

hipError_t rdyn_launch_regressor_tsqr_wide(int n_joints, const RdynLdsGramArgs& a, int blocks, double* workspace, double* R, int accumulate, hipStream_t st)
{
  const int n1 = 10 * n_joints + a.n_comp_cols + 1;
  const size_t lds = rdyn_regressor_tsqr_wide_lds_bytes(n1, a.n_active);
  if (lds == 0 || blocks < 1) return hipErrorInvalidValue;
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in((const void*)k_regressor_tsqr_wide, attr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_regressor_tsqr_wide, dim3(blocks), dim3(NTW), lds, st, a, n_joints, n1, workspace);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return wide_tree(workspace, blocks, workspace + (size_t)blocks * n1 * n1, R, n1, accumulate ? R : nullptr, a.run_flag, st);
}

hipError_t rdyn_launch_tsqr_wide_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                      int accumulate, const int* run_flag, hipStream_t st)
{
  const int n1 = n_cols + (b ? 1 : 0);
  const int rb = wide_rows_per_block(n1);
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols() || rb == 0 || blocks < 1) return hipErrorInvalidValue;
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in((const void*)k_tsqr_wide_rows, attr);
  if (e != hipSuccess) return e;
  // whole blocks per workgroup
  const int64_t n_blk = (rows + rb - 1) / rb;
  const int64_t per = (n_blk + blocks - 1) / blocks;
  const int used = (int)((n_blk + per - 1) / per);
  hipLaunchKernelGGL(k_tsqr_wide_rows, dim3(used), dim3(NTW), tri_bytes(n1) + (size_t)n1 * (rb + 4) * 8, st, A, b, rows, lda, n_cols, 0, (int64_t)0, rb, per * rb,
                     workspace, (const double*)nullptr, run_flag);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return wide_tree(workspace, used, workspace + (size_t)blocks * n1 * n1, R, n1, accumulate ? R : nullptr, run_flag, st);
}
