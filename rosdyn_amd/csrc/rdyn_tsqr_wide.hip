// rdyn_tsqr_wide.hip -- Householder R factor for the shapes the register-resident folds of rdyn_tsqr.hip cannot hold
// (no reference counterpart: the identification step lives outside rosdyn_core, /root/reference/README.md:15).
//
// rdyn_tsqr.hip keeps the running factor AND the row block in the registers of one wave: at most 64 columns for a device matrix,
// 7 chain joints (71 columns) for the fused sweep, 6 joints once friction / spring columns ride along.  Wider problems --
//   * [Y | C | tau_meas] of a 7-joint arm (friction_polynomial1.h:126, ideal_spring.h:64 columns beside getRegressor): 86 columns,
//   * a caller's materialised matrix of up to 111 columns + right-hand side (what rdyn_gram already takes)
// -- run here, with the factor in LDS: one workgroup keeps its running R PACKED in LDS (column j holds its j + 1 entries, 51 KB at
// 112 columns) and folds row block after row block into it, R <- qr([R ; block]).  The BLOCK lives in registers, spread over the
// workgroup: four threads per column, each holding a quarter of the column's rows (<= 32 doubles); a column step publishes column k
// (1 KB) through LDS, every thread reads its quarter of it, forms its partial dot, the four quarters meet by two DPP shuffles and the
// reflection is applied in registers: one barrier per step.  (The first version kept the block in LDS: 2.5 x rows x columns x 8 bytes
// of LDS traffic per step = 3 us per step at 128 rows x 86 columns -- the LDS pipe of the CU, not the dependent chain, was the
// bound; 26 ms for 1e6 rows x 112 columns.)  A block is
//   k_regressor_tsqr_wide   the 16-sample tile the workgroup's first wave has just swept (the row-pair sweeper of rdyn_duo_gram.hip /
//                           rdyn_tsqr.hip writing a RECTANGULAR tile: every column 16 n rows, the structural zeros stored), or
//   k_tsqr_wide_rows        up to 128 rows of a column-major device matrix -- also the tree: the per-workgroup factors are folded
//                           four at a time as the rows of a stacked matrix, in a fixed order (bitwise reproducible).
// A step is a dependent chain (LDS round trip, one square root, one division, 2 x 32 fmas) whatever the block holds.  This is the slow, unconditionally
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

constexpr int RPT_MAX = 32;       // rows of a block per thread: blocks of at most 128 rows
constexpr int PUB = 4 * RPT_MAX + 8;  // doubles of one publish buffer: the column + its sum of squares

// sum over the four lanes of a quad, in every lane: two DPP quad_perm moves per step (no LDS crossbar)
__device__ __forceinline__ double quad_sum(double x)
{
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, false);
    x += __hiloint2double(hi, lo);
  }
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x4E, 0xF, 0xF, false);    // quad_perm [2, 3, 0, 1]
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x4E, 0xF, 0xF, false);
  return x + __hiloint2double(hi, lo);
}

// the four owner lanes of a column write it, and its sum of squares, into a publish buffer
__device__ __forceinline__ void publish_column(double* pn, const double (&y)[RPT_MAX], int rpt, int q)
{
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int i = 0; i < RPT_MAX; i += 4)
  {
    if (i < rpt)  // rpt is a multiple of 4
    {
      pn[q * rpt + i] = y[i];
      pn[q * rpt + i + 1] = y[i + 1];
      pn[q * rpt + i + 2] = y[i + 2];
      pn[q * rpt + i + 3] = y[i + 3];
      s0 = fma(y[i], y[i], s0);
      s1 = fma(y[i + 1], y[i + 1], s1);
      s2 = fma(y[i + 2], y[i + 2], s2);
      s3 = fma(y[i + 3], y[i + 3], s3);
    }
  }
  const double sigma = quad_sum((s0 + s1) + (s2 + s3));
  if (q == 0) pn[4 * RPT_MAX] = sigma;
}

// R <- qr([R ; block]).  R packed in LDS (Rp[tri_off(j) + i], i <= j).  The block: thread (j = tid >> 2, q = tid & 3) holds rows
// q * rpt .. (q + 1) * rpt - 1 of column j in y[0 .. rpt) (rpt a multiple of 4; threads with j >= n1 hold nothing and only take part
// in the barriers); which rows those are is irrelevant as long as every column uses the same partition.  pub: 2 * PUB doubles of LDS.
// Column step k: the owners of column k have published it together with |y_k|^2 (no wave-wide reduction in the chain); every thread
// reads its quarter, forms its partial dot in four independent chains, the quarters meet by DPP, the reflection is applied in
// registers, the owners of column k + 1 publish: one barrier per step.
// The block is destroyed.  Ends with a barrier.  A column that is exactly zero (or rounding residue below 1e-140) in the block is passed.
__device__ __forceinline__ void wide_fold(double* Rp, double* pub, double (&y)[RPT_MAX], int rpt, int n1, int tid)
{
  const int j = tid >> 2, q = tid & 3;
  if (j == 0) publish_column(pub, y, rpt, q);
  __syncthreads();
  for (int k = 0; k < n1; ++k)
  {
    const double* const pk = pub + (k & 1) * PUB;
    const bool on = j > k && j < n1;
    double* const rjk = Rp + tri_off(on ? j : k) + k;
    const double rkj = *rjk;                      // R(k, j) (R(k, k) for the lanes that hold no column to the right)
    const double sigma = pk[4 * RPT_MAX];
    const double alpha = Rp[tri_off(k) + k];
    double beta = alpha;
    if (sigma > 1e-280)  // workgroup-uniform
    {
      const double norm = sqrt(fma(alpha, alpha, sigma));
      beta = alpha > 0.0 ? -norm : norm;
      const double v0 = alpha - beta, scale = 2.0 / fma(v0, v0, sigma);
      double yk[RPT_MAX];
      double d0 = (on && q == 0) ? v0 * rkj : 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
      for (int i = 0; i < RPT_MAX; i += 4)
        if (i < rpt)
        {
          yk[i] = pk[q * rpt + i];
          yk[i + 1] = pk[q * rpt + i + 1];
          yk[i + 2] = pk[q * rpt + i + 2];
          yk[i + 3] = pk[q * rpt + i + 3];
          d0 = fma(yk[i], y[i], d0);
          d1 = fma(yk[i + 1], y[i + 1], d1);
          d2 = fma(yk[i + 2], y[i + 2], d2);
          d3 = fma(yk[i + 3], y[i + 3], d3);
        }
      const double d = quad_sum((d0 + d1) + (d2 + d3));
      const double f = on ? scale * d : 0.0;
      if (on && q == 0) *rjk = fma(-f, v0, rkj);
#pragma unroll
      for (int i = 0; i < RPT_MAX; ++i)
        if (i < rpt) y[i] = fma(-f, yk[i], y[i]);
    }
    // the next column goes out (into the buffer nobody has read since the barrier before last)
    if (j == k + 1 && j < n1) publish_column(pub + ((k + 1) & 1) * PUB, y, rpt, q);
    __syncthreads();
    if (tid == 0) Rp[tri_off(k) + k] = beta;  // (behind the barrier: everybody has read alpha)
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
  double* const pub = (double*)(lds_raw + (((size_t)tri * 8 + 255) & ~(size_t)255));
  char* const tile = (char*)(pub + 2 * PUB);
  const int n = fa.n_active, rpt = 4 * n, ldb = fa.lds_stride[0] / 8;  // 16 n rows: 4 n per thread
  for (int i = tid; i < tri; i += NTW) Rp[i] = 0.0;
  const int s_loc = lane >> 2, k = lane & 3;
  const int r0 = k, r1 = k + 4;
    RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
  int fB = n_joints;
  for (int f = n_joints - 1; f >= 0; --f)
    if (fa.lds_m[f] >= 5) fB = f;
  const int64_t t_mul = fa.tile_stride > 1 ? fa.tile_stride : 1;
  const int64_t n_tiles = ((fa.n_samples + 15) / 16 + t_mul - 1) / t_mul;
  // the structural zeros of the rectangular tile are written once: the sweep stores the same positions for every tile, and the fold
  // works on a register copy
  for (int i = tid; i < n1 * ldb; i += NTW) ((double*)tile)[i] = 0.0;
  __syncthreads();
  for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x)
  {
    if (wave == 0)
    {
      int64_t sx = tl * t_mul * 16 + s_loc;
      const bool valid = sx < fa.n_samples;
      if (!valid) sx = fa.n_samples - 1;
      const int64_t o = sx * fa.in_ss;
      double qa = 0.0, dqa = 0.0, ddqa = 0.0, qb = 0.0, dqb = 0.0, ddqb = 0.0, tb0 = 0.0, tb1 = 0.0;
      if (fa.bcol)
      {
        if (r0 < n) tb0 = fa.bcol[o + in_oa];
        if (r1 < n) tb1 = fa.bcol[o + in_ob];
      }
      if (k < n)
      {
        qa = fa.q[o + in_oa];
        dqa = fa.dq[o + in_oa];
        ddqa = fa.ddq[o + in_oa];
      }
      if (k + 4 < n)
      {
        qb = fa.q[o + in_ob];
        dqb = fa.dq[o + in_ob];
        ddqb = fa.ddq[o + in_ob];
      }
      if (!valid) tb0 = tb1 = 0.0;
      const int m0idx = valid ? r0 : -2, m1idx = valid ? r1 : -2;
      double sna, csa, snb, csb;
      rdyn_sincos(qa, &sna, &csa);
      rdyn_sincos(qb, &snb, &csb);
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
    double y[RPT_MAX];
    {
      const int j = tid >> 2, q = tid & 3;
      const double* const col = (const double*)tile + (size_t)(j < n1 ? j : 0) * ldb + q * rpt;
#pragma unroll
      for (int i = 0; i < RPT_MAX; ++i) y[i] = (i < rpt && j < n1) ? col[i] : 0.0;
    }
    wide_fold(Rp, pub, y, rpt, n1, tid);
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
                                                        int seg_rows, int64_t seg_stride, int64_t rows_per_wg, double* __restrict__ out,
                                                        const double* __restrict__ extra, const int* __restrict__ run_flag)
{
  constexpr int RB = 4 * RPT_MAX;  // rows per block
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
  const int n1 = n_cols + (b ? 1 : 0), tri = tri_off(n1);
  double* const Rp = (double*)lds_raw;
  double* const pub = (double*)(lds_raw + (((size_t)tri * 8 + 255) & ~(size_t)255));
  int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t row_hi = row_lo + rows_per_wg < rows ? row_lo + rows_per_wg : rows;
  if (seg_rows == n1 && row_lo % n1 == 0 && row_lo + n1 <= row_hi && !b)
  {
    // tree levels: the workgroup's first factor IS a running factor -- taken as it stands instead of folded into zeros (a fold is
    // n1 dependent steps whatever its rows hold: 12 -> 8 folds on the way from 125 leaves to the result at 78 columns)
    const double* const F = A + (row_lo / n1) * seg_stride;
    for (int e = tid; e < n1 * n1; e += NTW)
    {
      const int i = e % n1, jj = e / n1;
      if (i <= jj) Rp[tri_off(jj) + i] = F[(int64_t)jj * lda + i];
    }
    row_lo += n1;
  }
  else
    for (int i = tid; i < tri; i += NTW) Rp[i] = 0.0;
  __syncthreads();
  for (int64_t r0 = row_lo; r0 < row_hi; r0 += RB)
  {
    // my quarter of column j: rows r0 + q * 32 .. + 31 (256 contiguous bytes of a column-major matrix)
    double y[RPT_MAX];
    const int64_t g0 = r0 + q * RPT_MAX;
#pragma unroll
    for (int i = 0; i < RPT_MAX; ++i)
    {
      const int64_t gr = g0 + i;
      double v = 0.0;
      if (j < n1 && gr < row_hi)
      {
        if (j >= n_cols)
          v = b[gr];
        else if (seg_rows > 0)
        {
          const int64_t g = gr / seg_rows;
          v = A[g * seg_stride + (int64_t)j * lda + (gr - g * seg_rows)];
        }
        else
          v = A[(int64_t)j * lda + gr];
      }
      y[i] = v;
    }
    wide_fold(Rp, pub, y, RPT_MAX, n1, tid);
  }
  if (extra && blockIdx.x == 0)
  {
    for (int r0 = 0; r0 < n1; r0 += RB)
    {
      double y[RPT_MAX];
#pragma unroll
      for (int i = 0; i < RPT_MAX; ++i)
      {
        const int gr = r0 + q * RPT_MAX + i;
        y[i] = (j < n1 && gr < n1 && gr <= j) ? extra[(int64_t)j * n1 + gr] : 0.0;
      }
      wide_fold(Rp, pub, y, RPT_MAX, n1, tid);
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
constexpr size_t kPubBytes = 2 * PUB * 8;
constexpr int kRowsPerBlock = 4 * RPT_MAX;

int wide_tree_fan(int n1)
{
  const int f = 1 + 2 * kRowsPerBlock / n1;
  return f < 3 ? 3 : (f > 8 ? 8 : f);
}
// count factors (n1 x n1, column-major, contiguous) at `in` -> R, `fan` per workgroup and level; scratch: room for the levels' factors
// (count / 3 + count / 9 + ... < count / 2 + one per level)
hipError_t wide_tree(const double* in, int count, double* scratch, double* R, int n1, const double* extra, const int* run_flag, hipStream_t st,
                     int64_t in_stride = 0)
{
  int64_t stride = in_stride > 0 ? in_stride : (int64_t)n1 * n1;  // doubles between the factors of the first level
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in((const void*)k_tsqr_wide_rows, attr);
  if (e != hipSuccess) return e;
  const size_t lds = tri_bytes(n1) + kPubBytes;
  // a level costs ceil((fan - 1) n1 / 128) folds of n1 dependent steps each (the first factor is taken as it stands): the widest fan
  // whose other factors fill two row blocks -- 3 at 112 columns, 4 at 78, 5 at 61, at most 8
  const int fan = wide_tree_fan(n1);
  while (true)
  {
    const int nout = (count + fan - 1) / fan;
    const bool last = nout <= 1;
    hipLaunchKernelGGL(k_tsqr_wide_rows, dim3(last ? 1 : nout), dim3(NTW), lds, st, in, (const double*)nullptr, (int64_t)count * n1, (int64_t)n1, n1, n1,
                       stride, (int64_t)fan * n1, last ? R : scratch, last ? extra : (const double*)nullptr, run_flag);
    e = hipGetLastError();
    if (e != hipSuccess || last) return e;
    in = scratch;
    stride = (int64_t)n1 * n1;
    count = nout;
    scratch = scratch + (size_t)nout * n1 * n1;
  }
}
}  // namespace

// widest factor the LDS-resident folds serve (right-hand side included)
int rdyn_tsqr_wide_max_cols() { return 112; }

// LDS of a k_regressor_tsqr_wide launch (0: the tile does not fit beside the factor)
size_t rdyn_regressor_tsqr_wide_lds_bytes(int n1, int n_active)
{
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols() || n_active < 1 || n_active > 8) return 0;
  const size_t bytes = tri_bytes(n1) + kPubBytes + (size_t)n1 * (16 * n_active + 4) * 8;
  return bytes <= kWideLdsBudget ? bytes : 0;
}

// doubles of workspace: the leaves' factors + the tree's intermediate level
size_t rdyn_tsqr_wide_workspace_doubles(int n1, int blocks) { return (size_t)(blocks + (blocks + 1) / 2 + 12) * n1 * n1; }

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

// R <- the fold of `count` upper-triangular n1 x n1 factors (column-major, `stride` doubles apart), plus R itself when accumulating;
// fixed order.  scratch: rdyn_tsqr_wide_workspace_doubles(n1, count) doubles.
hipError_t rdyn_launch_tsqr_fold_factors(const double* factors, int count, int64_t stride, int n1, double* scratch, double* R, int accumulate, hipStream_t st)
{
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols() || count < 1 || stride < (int64_t)n1 * n1) return hipErrorInvalidValue;
  return wide_tree(factors, count, scratch, R, n1, accumulate ? R : nullptr, nullptr, st, stride);
}

hipError_t rdyn_launch_tsqr_wide_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                      int accumulate, const int* run_flag, hipStream_t st)
{
  const int n1 = n_cols + (b ? 1 : 0);
  const int rb = kRowsPerBlock;
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols() || blocks < 1) return hipErrorInvalidValue;
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in((const void*)k_tsqr_wide_rows, attr);
  if (e != hipSuccess) return e;
  // whole blocks per workgroup
  const int64_t n_blk = (rows + rb - 1) / rb;
  const int64_t per = (n_blk + blocks - 1) / blocks;
  const int used = (int)((n_blk + per - 1) / per);
  hipLaunchKernelGGL(k_tsqr_wide_rows, dim3(used), dim3(NTW), tri_bytes(n1) + kPubBytes, st, A, b, rows, lda, n_cols, 0, (int64_t)0, per * rb,
                     workspace, (const double*)nullptr, run_flag);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return wide_tree(workspace, used, workspace + (size_t)blocks * n1 * n1, R, n1, accumulate ? R : nullptr, run_flag, st);
}
