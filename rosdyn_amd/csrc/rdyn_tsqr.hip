// rdyn_tsqr.hip -- tall-skinny QR of the stacked regressor [A | tau_meas] WITHOUT forming A'A (BASELINE.json configs[2]:
// "regressor + TSQR Gram").  The Gram route (rdyn_duo_gram.hip -> pivoted Cholesky) squares the condition number: fine for
// excitation trajectories, useless beyond cond(A) ~ 1e8.  Here the R factor itself is accumulated:
//
//   leaf     every wave keeps a running upper-triangular R in registers and folds row blocks into it by Householder reflections:
//            R <- qr([R ; block]).  A block is the LDS tile of 16 samples that the wave's own row-pair sweep has just produced
//            (rdyn_regressor_tsqr; packed column-major tile of rdyn_lds_gram.hip: the columns of link f only hold the rows of the
//            joints that can be non-zero there) or 64 rows of a device matrix (rdyn_tsqr).  Block and factor are spread over the
//            wave in two dimensions (16 column slots x 4 row groups, tsqr_fold2d below); a column step exchanges one column
//            through LDS and sums the row groups with v_permlane16/32_swap.  A wave never touches another wave's data in the loop.
//   block    the four waves of a workgroup fold their factors 1 -> 0, 3 -> 2, 2 -> 0 through LDS before anything is written.
//   tree     k_tsqr_combine folds two factors into one per wave, level by level in a fixed order (bitwise reproducible).
//   ranks    the (P + 1)^2 factor of every rank is all-gathered and folded on the host (rdyn_tsqr_combine_host): same payload
//            as the Gram all-reduce.
// Result: R1 = [R d; 0 rho] with A = Q R, d = Q' tau_meas, rho = |residual| -- solve with rdyn_solve_r_factor.
// Cost (1e6 samples, n = 6): 3.2 ms = 5 x the Gram kernel, of which 0.4 ms are the block + tree levels (10 dependent folds).  A fold
// is 61 dependent column steps of ~1 400 cycles: LDS round trip 40 %, the fmas 35 %, row-group sums, sqrt and divide the rest --
// a latency chain at one wave per SIMD (300+ VGPRs), not a throughput bound.  The Gram stays the default; this is the robust path.
// Round 2's first version (lane = column, block re-read from LDS twice per step) ran 8.7 ms at the LDS roofline.
// fp64 VALU + LDS only (Householder updates are rank-1: nothing for the matrix cores at this width).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <utility>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_duo_common.h"

#define DUO_BARRIER()  // the link body shared with rdyn_duo_gram.hip synchronises with its consumer wave there; not here

namespace
{
typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));

__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------- leaf update, 2-D register-resident (round 2b)
// The lane = column update above moves every block element through LDS twice per column step (|y_k|^2 / y_k . y_j, then the
// reflection): 1 350 LDS cycles per step and wave, and the LDS pipe is shared by the CU -- 8.7 ms at config 2 was the LDS roofline,
// with the fp64 pipes a quarter busy.  The leaves therefore keep the block in REGISTERS, spread over the wave in two dimensions:
//   lane = (cs = lane & 15, rg = lane >> 4):  column j lives in column slot (j >> 4) of the 16 lanes with cs == (j & 15); row slot r
//   of the block is 16 rows, row group rg owns rows 4 rg .. 4 rg + 3 of every slot (sweep tiles: slot = joint row, rows = samples).
//   R[k][j] lives in lane (cs = j & 15, rg = k & 3), register Rr[j >> 4][k >> 2].
// Column step k: the four lanes that own y_k publish it (<= 1 KB) together with row k of R, every lane reads the slice of its row
// group, forms the partial dots of its <= NCI columns, the four row groups are summed with two v_permlane*_swap exchanges (gfx950:
// no LDS), and the reflection is applied in registers.  Work per step and lane: rows/4 * (columns right of k)/16 fmas instead of
// `rows` fmas for the active lanes only -- and 2-3 LDS instructions per row slot instead of 6 per row pair.
// The step loop is unrolled over (k >> 4, (k >> 2) & 3) so that every register index is static and rolled over k & 3.
__device__ __forceinline__ double rowgroup_sum(double x)
{
  // v_permlane16_swap a, b: rows 1, 3 of a <-> rows 0, 2 of b (row = 16 lanes); with a = b = x: a' = [x0 x0 x2 x2], b' = [x1 x1 x3 x3]
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  const unsigned lo2 = (unsigned)__double2loint(s), hi2 = (unsigned)__double2hiint(s);
  const auto c = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);  // upper half of a <-> lower half of b
  const auto d = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
  return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}

// RMaxOf::of(c): static bound of the row slots the columns of column slot c can hold (non-decreasing in c); the slots a column does
// not store are zeros in Y and stay zeros (a column never stores fewer slots than a column to its left), so the steps run over
// RMaxOf::of(k >> 4) slots without run-time row counts: ~20 % more fmas than the exact counts of a sweep tile, no selects.
// buf: >= (RM * 16 + NCI * 16) doubles of LDS, wave-private.
template <int NC, int RM, class RMaxOf, int KCI, int KQ>
__device__ __forceinline__ void tsqr_fold2d_steps(double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], double (&Y)[(NC + 15) / 16][RM][4], char* buf, int lane)
{
  constexpr int NCI = (NC + 15) / 16;
  constexpr int LEFT = NC - (16 * KCI + 4 * KQ), KR = LEFT < 4 ? LEFT : 4;  // steps 16 KCI + 4 KQ + (0 .. KR - 1)
  if constexpr (KR > 0)
  {
    const int cs = lane & 15, rg = lane >> 4;
    char* const pv = buf + rg * 32;       // my row group's slice of the published column: [r][16 rows]
    char* const rrow = buf + RM * 128;    // row k of R: [NCI * 16]
#pragma unroll 1
  for (int kr = 0; kr < KR; ++kr)
  {
    const int kcs = 4 * KQ + kr, k = 16 * KCI + kcs;
    // ---- publish y_k (its four owner lanes) and row k of R (the 16 lanes of row group k & 3)
    if (cs == kcs)
    {
#pragma unroll
      for (int r = 0; r < RM; ++r)
        if (r < RMaxOf::of(KCI))
        {
          d2a v0, v1;
          v0.x = Y[KCI][r][0]; v0.y = Y[KCI][r][1]; v1.x = Y[KCI][r][2]; v1.y = Y[KCI][r][3];
          *(d2a*)(pv + r * 128) = v0;
          *(d2a*)(pv + r * 128 + 16) = v1;
        }
    }
    if (rg == kr)
    {
#pragma unroll
      for (int c = KCI; c < NCI; ++c) *(double*)(rrow + (16 * c + cs) * 8) = Rr[c][4 * KCI + KQ];
    }
    wave_lds_fence();
    // ---- my row group's slice of y_k, my columns' entries of row k
    double yk[RM][4], rk[NCI], dp[NCI];
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if (r < RMaxOf::of(KCI))
      {
        const d2a v0 = *(const d2a*)(pv + r * 128), v1 = *(const d2a*)(pv + r * 128 + 16);
        yk[r][0] = v0.x; yk[r][1] = v0.y; yk[r][2] = v1.x; yk[r][3] = v1.y;
      }
#pragma unroll
    for (int c = KCI; c < NCI; ++c) rk[c] = *(const double*)(rrow + (16 * c + cs) * 8);
    const double alpha = *(const double*)(rrow + k * 8);
#pragma unroll
    for (int c = KCI; c < NCI; ++c) dp[c] = 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if (r < RMaxOf::of(KCI))
      {
#pragma unroll
        for (int c = KCI; c < NCI; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) dp[c] = fma(yk[r][i], Y[c][r][i], dp[c]);
      }
#pragma unroll
    for (int c = KCI; c < NCI; ++c) dp[c] = rowgroup_sum(dp[c]);
    // |y_k|^2 is the dot of column k with itself: lane kcs (row group 0) holds it in dp[KCI]
    const double sigma = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dp[KCI]), kcs), __builtin_amdgcn_readlane(__double2loint(dp[KCI]), kcs));
    // a column that is roundoff residue of earlier reflections (more columns than rows so far) shrinks by ~1e-17 per step; past
    // ~1e-140 its square underflows, 2 / (v0^2 + sigma) overflows and the factor fills with NaN.  Such a column carries no
    // information: it is passed like an exactly zero one (its entries stay below 1e-140 and leave with the block).
    if (sigma > 1e-280)  // wave-uniform
    {
      const double norm = sqrt(fma(alpha, alpha, sigma));
      const double beta = alpha > 0.0 ? -norm : norm;
      const double v0 = alpha - beta;
      const double scale = 2.0 / fma(v0, v0, sigma);
#pragma unroll
      for (int c = KCI; c < NCI; ++c)
      {
        const int j = 16 * c + cs;
        const double f = (j > k && j < NC) ? scale * fma(v0, rk[c], dp[c]) : 0.0;
        if (rg == kr) Rr[c][4 * KCI + KQ] = j == k ? beta : fma(-f, v0, rk[c]);
#pragma unroll
        for (int r = 0; r < RM; ++r)
          if (r < RMaxOf::of(KCI))
          {
#pragma unroll
            for (int i = 0; i < 4; ++i) Y[c][r][i] = fma(-f, yk[r][i], Y[c][r][i]);
          }
      }
    }
    wave_lds_fence();  // the next step publishes into the same buffer
  }
  }
}

template <int NC, int RM, class RMaxOf, int... I>
__device__ __forceinline__ void tsqr_fold2d_seq(double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], double (&Y)[(NC + 15) / 16][RM][4], char* buf, int lane,
                                                std::integer_sequence<int, I...>)
{
  (tsqr_fold2d_steps<NC, RM, RMaxOf, I / 4, I % 4>(Rr, Y, buf, lane), ...);
}

template <int NC, int RM, class RMaxOf>
__device__ __forceinline__ void tsqr_fold2d(double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], double (&Y)[(NC + 15) / 16][RM][4], char* buf, int lane)
{
  tsqr_fold2d_seq<NC, RM, RMaxOf>(Rr, Y, buf, lane, std::make_integer_sequence<int, 4 * ((NC + 15) / 16)>{});
}

// R[k][j] of the 2-D distribution -> n_cols x n_cols factor, column-major with leading dimension ld (zeros below the diagonal)
template <int NC, class Ptr>
__device__ __forceinline__ void store_factor2d(const double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], Ptr out, int ld, int n_cols, int lane)
{
  const int cs = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int c = 0; c < (NC + 15) / 16; ++c)
#pragma unroll
    for (int kk = 0; kk < (NC + 3) / 4; ++kk)
    {
      const int j = 16 * c + cs, k = 4 * kk + rg;
      if (j < n_cols && k < n_cols) out[j * ld + k] = k <= j ? Rr[c][kk] : 0.0;
    }
}

// the running factor from memory (inverse of store_factor2d)
template <int NC, class Ptr>
__device__ __forceinline__ void load_factor2d(double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], Ptr in, int ld, int n_cols, int lane)
{
  const int cs = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int c = 0; c < (NC + 15) / 16; ++c)
#pragma unroll
    for (int kk = 0; kk < (NC + 3) / 4; ++kk)
    {
      const int j = 16 * c + cs, k = 4 * kk + rg;
      Rr[c][kk] = (j < n_cols && k <= j) ? in[j * ld + k] : 0.0;
    }
}

// an upper-triangular factor as the BLOCK of a fold: column 16 c + cs stores rows 0 .. 16 c + cs, i.e. at most c + 1 row slots
struct TriRowSlots
{
  static constexpr int of(int c) { return c + 1; }
};
template <int NC, class Ptr>
__device__ __forceinline__ void load_tri_block(double (&Y)[(NC + 15) / 16][(NC + 15) / 16][4], Ptr f, int ld, int n_cols, int lane)
{
  const int cs = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int c = 0; c < (NC + 15) / 16; ++c)
#pragma unroll
    for (int r = 0; r <= c; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        const int j = 16 * c + cs, row = 16 * r + 4 * rg + i;
        Y[c][r][i] = (j < n_cols && row <= j) ? f[j * ld + row] : 0.0;
      }
}

// leading dimension of a factor parked in LDS: even (16-byte aligned row slices) and not a multiple of 256 bytes (banks)
constexpr int lds_factor_ld(int nc) { return ((nc + 1) & ~1) + ((((nc + 1) & ~1) * 8) % 256 == 0 ? 2 : 0); }

// the four running factors of a workgroup -> one (wave 0's), through LDS: 1 -> 0 and 3 -> 2, then 2 -> 0, a fixed order.
// region(w): LDS of wave w, at least NC * lds_factor_ld(NC) doubles from region(0) / region(2) on (two waves' regions are contiguous)
template <int NC, class Region>
__device__ __forceinline__ void block_combine(double (&Rr)[(NC + 15) / 16][(NC + 3) / 4], Region region, int wave, int lane)
{
  constexpr int NCI = (NC + 15) / 16, LD = lds_factor_ld(NC);
#pragma unroll 1
  for (int level = 0; level < 2; ++level)
  {
    const int sender = level == 0 ? (wave & 1) : (wave == 2), receiver = level == 0 ? !(wave & 1) : (wave == 0);
    __syncthreads();  // the regions are free (tile loop / previous level done)
    if (sender) store_factor2d<NC>(Rr, (double*)region(level == 0 ? wave - 1 : 0), LD, NC, lane);
    __syncthreads();
    if (receiver)
    {
      char* const reg = region(wave);
      double Y[NCI][NCI][4];
      load_tri_block<NC>(Y, (const double*)reg, LD, NC, lane);
      wave_lds_fence();
      tsqr_fold2d<NC, NCI, TriRowSlots>(Rr, Y, reg, lane);
    }
  }
}

// ---------------------------------------------------------------- leaf: regressor rows from the wave's own sweep
template <int NJ>
struct SweepRowSlots  // columns 16 c .. 16 c + 15 belong to links <= (16 c + 15) / 10, whose columns store <= link + 1 joint rows
{
  static constexpr int of(int c) { return (16 * c + 15) / 10 + 1 < NJ ? (16 * c + 15) / 10 + 1 : NJ; }
};

// XC: 16-column slots for the component columns of rdyn_identification_tsqr ([Y | C | tau_meas]); 0: [Y | tau_meas] with exact widths.
// With components the factor is NC = 16 (ceil((P + 1) / 16) + XC) wide; columns P + K + 1 .. NC - 1 are zero (their steps find
// sigma = 0 and pass).
template <int NJ, int XC>
__global__ __launch_bounds__(256) void k_regressor_tsqr(const RdynLdsGramArgs fa, double* __restrict__ factors)
{
  constexpr int P = 10 * NJ, NC = XC == 0 ? P + 1 : 16 * ((P + 1 + 15) / 16 + XC), NCI = (NC + 15) / 16, NK = (NC + 3) / 4;
  const int K = XC > 0 ? fa.n_comp_cols : 0;  // component columns sit between the regressor and tau_meas
  constexpr bool DIRECT = false, ALLREV = false;
  if (fa.run_flag && *fa.run_flag == 0) return;  // stand-by call, not needed (uniform: every wave leaves)
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  ChainPtr c = as_const(fa.chain);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const tile = lds_raw + (size_t)wave * fa.tile_bytes;
  const int n = fa.n_active;
  const int s_loc = lane >> 2, k = lane & 3;
  const int r0 = k, r1 = k + 4;
    RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
  int fB = NJ;
  for (int f = NJ - 1; f >= 0; --f)
    if (fa.lds_m[f] >= 5) fB = f;
  // tile column p: byte offset of its row slot 0 and the row slots [lo, hi) it stores (joint rows; 16 samples each)
  auto col_of = [&](int p, int& base, int& lo, int& hi) {
    lo = 0;
    if (p < P)
    {
      const int f = p / 10;
      base = fa.lds_off[f] + (p - 10 * f) * fa.lds_stride[f];
      hi = fa.lds_m[f];
    }
    else if (p < P + K)
    {
      lo = fa.comp_col_row[p - P];  // a component column is one row slot: its own joint's
      hi = lo + 1;
      base = fa.lds_off_c + (p - P) * fa.comp_stride - lo * 128;
    }
    else if (p == P + K)
    {
      base = fa.lds_off_b;
      hi = n;
    }
    else
    {
      base = 0;
      hi = 0;  // padding column
    }
  };
  // my columns of the tile (2-D distribution of tsqr_fold2d): column 16 c + cs, rows 4 rg .. 4 rg + 3 of every slot
  const int cs = lane & 15, rg = lane >> 4;
  int yb[NCI], ylo[NCI], ys[NCI];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
  {
    const int j = 16 * ci + cs;
    col_of(j < NC ? j : NC - 1, yb[ci], ylo[ci], ys[ci]);
    yb[ci] += rg * 32;
    if (j >= NC) ys[ci] = 0;
  }
  double Rr[NCI][NK];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
    for (int i = 0; i < NK; ++i) Rr[ci][i] = 0.0;

  // tile_stride > 1: only every tile_stride-th tile (the subsample pass of rdyn_cholqr.hip)
  const int64_t t_mul = fa.tile_stride > 1 ? fa.tile_stride : 1;
  const int64_t n_tiles = ((fa.n_samples + 15) / 16 + t_mul - 1) / t_mul;
  const int64_t t_first = (int64_t)blockIdx.x * 4 + wave, t_step = (int64_t)gridDim.x * 4;
  for (int64_t tl = t_first; tl < n_tiles; tl += t_step)
  {
    // ---------------- sweep (row-pair lanes, as the sweeper of rdyn_duo_gram.hip): my sample's rows k and k + 4 -> LDS tile
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
    for (int f = 0; f < NJ; ++f)
    {
#include "rdyn_duo_link_body.inc"
    }
    if (XC > 0)
    {
#include "rdyn_duo_comp_cols.inc"
    }
    {
      char* const lb = tile + fa.lds_off_b + s_loc * 8;
      if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
      if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
    }
    wave_lds_fence();
    // ---------------- the tile moves into registers and is folded into the running factor; its LDS is the fold's exchange buffer
    double Y[NCI][NJ][4];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int r = 0; r < NJ; ++r)
        if (r < SweepRowSlots<NJ>::of(ci))
        {
          d2a v0 = {0.0, 0.0}, v1 = {0.0, 0.0};
          if (r < ys[ci] && r >= ylo[ci])
          {
            v0 = *(const d2a*)(tile + yb[ci] + r * 128);
            v1 = *(const d2a*)(tile + yb[ci] + r * 128 + 16);
          }
          Y[ci][r][0] = v0.x; Y[ci][r][1] = v0.y; Y[ci][r][2] = v1.x; Y[ci][r][3] = v1.y;
        }
    wave_lds_fence();
    tsqr_fold2d<NC, NJ, SweepRowSlots<NJ>>(Rr, Y, tile, lane);
  }
  // ---------------- the workgroup's four factors -> one
  // (two waves' regions hold one parked factor: NC * ld doubles; with component columns on a short chain a factor is larger than two
  // tiles, the launcher then sizes the workgroup's LDS for two factors)
  const size_t cstride = (size_t)fa.tile_bytes * 2 > (size_t)NC * lds_factor_ld(NC) * 8 ? (size_t)fa.tile_bytes : ((size_t)NC * lds_factor_ld(NC) * 8 + 63) / 64 * 32;
  block_combine<NC>(Rr, [&](int w) { return lds_raw + (size_t)w * cstride; }, wave, lane);
  if (wave == 0) store_factor2d<NC>(Rr, factors + (int64_t)blockIdx.x * (NC * NC), NC, NC, lane);
}

// ---------------------------------------------------------------- leaf: 64-row blocks of a column-major device matrix
template <int SLOTS>
struct ConstRowSlots
{
  static constexpr int of(int) { return SLOTS; }
};

template <int NC>
__global__ __launch_bounds__(256) void k_tsqr_rows(const double* __restrict__ A, const double* __restrict__ b, int64_t rows, int64_t lda, int n_cols,
                                                   double* __restrict__ factors, const int* __restrict__ run_flag)
{
  if (run_flag && *run_flag == 0) return;  // stand-by call, not needed
  constexpr int TR = 64, SB = (TR + 2) * 8;  // block: NC columns of TR rows (+ pad), column-major
  constexpr int NCI = (NC + 15) / 16, NK = (NC + 3) / 4, RM = TR / 16;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const blk = lds_raw + (size_t)wave * (NC * SB);
  const int cs = lane & 15, rg = lane >> 4;
  double Rr[NCI][NK];
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
    for (int i = 0; i < NK; ++i) Rr[ci][i] = 0.0;
  const int nc1 = n_cols + (b ? 1 : 0);  // the right-hand side rides as one more column
  const int64_t n_blocks = (rows + TR - 1) / TR;
  const int64_t b_first = (int64_t)blockIdx.x * 4 + wave, b_step = (int64_t)gridDim.x * 4;
  for (int64_t bi = b_first; bi < n_blocks; bi += b_step)
  {
    // coalesced along the rows (512 B per column), transposed into the 2-D distribution through LDS
    const int64_t r = bi * TR + lane;
    for (int col = 0; col < nc1; ++col)
    {
      double v = 0.0;
      if (r < rows) v = col < n_cols ? A[(int64_t)col * lda + r] : b[r];
      *(double*)(blk + col * SB + lane * 8) = v;
    }
    wave_lds_fence();
    double Y[NCI][RM][4];
#pragma unroll
    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
      for (int rs = 0; rs < RM; ++rs)
      {
        d2a v0 = {0.0, 0.0}, v1 = {0.0, 0.0};
        const int j = 16 * ci + cs;
        if (j < nc1)  // the padding columns (nc1 .. NC - 1) are never loaded
        {
          v0 = *(const d2a*)(blk + j * SB + (rs * 16 + rg * 4) * 8);
          v1 = *(const d2a*)(blk + j * SB + (rs * 16 + rg * 4) * 8 + 16);
        }
        Y[ci][rs][0] = v0.x; Y[ci][rs][1] = v0.y; Y[ci][rs][2] = v1.x; Y[ci][rs][3] = v1.y;
      }
    wave_lds_fence();
    tsqr_fold2d<NC, RM, ConstRowSlots<RM>>(Rr, Y, blk, lane);
  }
  block_combine<NC>(Rr, [&](int w) { return lds_raw + (size_t)w * (NC * SB); }, wave, lane);
  if (wave == 0) store_factor2d<NC>(Rr, factors + (int64_t)blockIdx.x * (NC * NC), NC, NC, lane);
}

// ---------------------------------------------------------------- tree: every wave folds `fan` factors into one
// (2 : 1 by default -- a fold is 61 .. 71 dependent column steps whatever the block holds, so the DEPTH of the tree is its cost: 4 : 1
// levels fold three blocks one after the other per level.  A larger fan trades time for launches: the stand-by call of rdyn_api.cpp)
template <int NC>
__global__ __launch_bounds__(64) void k_tsqr_combine(const double* __restrict__ in, int count, int fan, double* __restrict__ out, int out_ld, int out_cols,
                                                     const double* __restrict__ extra /* one more factor (accumulate), or null */, int extra_ld,
                                                     const int* __restrict__ run_flag)
{
  constexpr int NCI = (NC + 15) / 16, NK = (NC + 3) / 4;
  if (run_flag && *run_flag == 0) return;
  __shared__ __attribute__((aligned(32))) char buf[2 * NCI * 128];
  const int lane = threadIdx.x;
  const int first = blockIdx.x * fan;
  double Rr[NCI][NK];
  load_factor2d<NC>(Rr, in + (int64_t)first * (NC * NC), NC, NC, lane);
  const int n_more = count - first - 1 < fan - 1 ? count - first - 1 : fan - 1;
  for (int t = 1; t <= n_more + (extra && blockIdx.x == 0 ? 1 : 0); ++t)
  {
    const bool is_extra = t > n_more;
    double Y[NCI][NCI][4];
    if (is_extra)
      load_tri_block<NC>(Y, extra, extra_ld, out_cols, lane);
    else
      load_tri_block<NC>(Y, in + (int64_t)(first + t) * (NC * NC), NC, NC, lane);
    tsqr_fold2d<NC, NCI, TriRowSlots>(Rr, Y, buf, lane);
  }
  store_factor2d<NC>(Rr, out + (int64_t)blockIdx.x * (NC * NC), out_ld, out_cols, lane);  // the last level is one wave: offset 0
}

template <class K>
hipError_t opt_in_lds(K kernel, std::atomic<uint64_t>& done)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  return hipSuccess;
}

// folds `count` NC x NC factors at `slab` down to one, written to R (ld_out x n_out, column-major); scratch = second slab region
template <int NC>
hipError_t combine_tree(double* slab, int count, double* scratch, double* R, int n_out, const double* extra, hipStream_t st, int fan = 2,
                        const int* run_flag = nullptr)
{
  double* in = slab;
  double* out = scratch;
  while (count > fan)
  {
    const int nout = (count + fan - 1) / fan;
    hipLaunchKernelGGL((k_tsqr_combine<NC>), dim3(nout), dim3(64), 0, st, in, count, fan, out, NC, NC, nullptr, 0, run_flag);
    count = nout;
    double* t = in;
    in = out;
    out = t;
  }
  // last level: straight into the caller's buffer (compact n_out x n_out), folding the caller's previous factor if accumulating
  hipLaunchKernelGGL((k_tsqr_combine<NC>), dim3(1), dim3(64), 0, st, in, count, fan, R, n_out, n_out, extra, n_out, run_flag);
  return hipGetLastError();
}

template <int NJ, int XC>
hipError_t launch_regressor_tsqr(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, double* slab, double* scratch, double* R, const double* extra,
                                 hipStream_t st, int fan)
{
  constexpr int NC = XC == 0 ? 10 * NJ + 1 : 16 * ((10 * NJ + 1 + 15) / 16 + XC);
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds(k_regressor_tsqr<NJ, XC>, attr);
  if (e != hipSuccess) return e;
  const size_t two_factors = 2 * (((size_t)NC * lds_factor_ld(NC) * 8 + 63) / 64 * 64);
  if (lds_bytes < two_factors) lds_bytes = two_factors;  // the block combine parks two factors side by side
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_regressor_tsqr<NJ, XC>), dim3(blocks), dim3(256), lds_bytes, st, a, slab);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return combine_tree<NC>(slab, blocks, scratch, R, 10 * NJ + (XC > 0 ? a.n_comp_cols : 0) + 1, extra, st, fan, a.run_flag);
}

template <int NC>
hipError_t launch_tsqr_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* slab, double* scratch, double* R,
                            const double* extra, hipStream_t st, const int* run_flag, int fan)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds(k_tsqr_rows<NC>, attr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tsqr_rows<NC>), dim3(blocks), dim3(256), (size_t)4 * NC * ((64 + 2) * 8), st, A, b, rows, lda, n_cols, slab, run_flag);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return combine_tree<NC>(slab, blocks, scratch, R, n_cols + (b ? 1 : 0), extra, st, fan, run_flag);
}
}  // namespace

// factors per launch and doubles of workspace (two slab regions: leaves + tree levels)
int rdyn_tsqr_padded_cols(int n_cols_with_rhs) { return n_cols_with_rhs <= 16 ? 16 : n_cols_with_rhs <= 32 ? 32 : n_cols_with_rhs <= 48 ? 48 : n_cols_with_rhs <= 64 ? 64 : 0; }
size_t rdyn_tsqr_workspace_doubles(int nc, int blocks) { return (size_t)(blocks + (blocks + 1) / 2 + 2) * nc * nc; }

// width of the factors a launch works with: 10 nJ + 1, or (with component columns) the next multiple of 16 plus one 16-column slot
int rdyn_regressor_tsqr_cols(int n_joints, int n_comp_cols)
{
  const int p1 = 10 * n_joints + 1;
  if (n_comp_cols <= 0) return p1;
  const int nc = 16 * ((p1 + 15) / 16 + 1);
  return (n_joints >= 2 && n_joints <= 6 && p1 + n_comp_cols <= nc) ? nc : 0;  // 7 joints + components: the fold would not fit the register file
}

hipError_t rdyn_launch_regressor_tsqr(int n_joints, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, double* workspace, double* R, int accumulate,
                                      hipStream_t st, int tree_fan)
{
  const int nc = rdyn_regressor_tsqr_cols(n_joints, a.n_comp_cols);
  if (nc == 0 || tree_fan < 2) return hipErrorInvalidValue;
  double* slab = workspace;
  double* scratch = workspace + (size_t)blocks * nc * nc;
  const double* extra = accumulate ? R : nullptr;
  if (a.n_comp_cols > 0)
    switch (n_joints)
    {
    case 2: return launch_regressor_tsqr<2, 1>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
    case 3: return launch_regressor_tsqr<3, 1>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
    case 4: return launch_regressor_tsqr<4, 1>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
    case 5: return launch_regressor_tsqr<5, 1>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
    case 6: return launch_regressor_tsqr<6, 1>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
    default: return hipErrorInvalidValue;
    }
  switch (n_joints)
  {
  case 2: return launch_regressor_tsqr<2, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  case 3: return launch_regressor_tsqr<3, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  case 4: return launch_regressor_tsqr<4, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  case 5: return launch_regressor_tsqr<5, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  case 6: return launch_regressor_tsqr<6, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  case 7: return launch_regressor_tsqr<7, 0>(a, blocks, lds_bytes, slab, scratch, R, extra, st, tree_fan);
  default: return hipErrorInvalidValue;
  }
}

hipError_t rdyn_launch_tsqr_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                 int accumulate, hipStream_t st, const int* run_flag, int tree_fan)
{
  const int nc = rdyn_tsqr_padded_cols(n_cols + (b ? 1 : 0));
  double* slab = workspace;
  double* scratch = workspace + (size_t)blocks * nc * nc;
  const double* extra = accumulate ? R : nullptr;
  switch (nc)
  {
  case 16: return launch_tsqr_rows<16>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st, run_flag, tree_fan);
  case 32: return launch_tsqr_rows<32>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st, run_flag, tree_fan);
  case 48: return launch_tsqr_rows<48>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st, run_flag, tree_fan);
  case 64: return launch_tsqr_rows<64>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st, run_flag, tree_fan);
  default: return hipErrorInvalidValue;
  }
}
