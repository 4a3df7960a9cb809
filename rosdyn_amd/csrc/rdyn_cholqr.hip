// rdyn_cholqr.hip -- the robust R factor of the stacked regressor [A | tau_meas] with the heavy pass on the fp64 MATRIX cores
// (BASELINE.json configs[2]: "regressor + TSQR Gram ..., MFMA path"; no reference counterpart: /root/reference/README.md:15).
//
// rdyn_tsqr.hip folds Householder reflections on the vector units: rank-1 updates, a ~1 400-cycle dependent chain per column step,
// 15.8 ms at config 3 (7 joints, 4e6 samples) with the matrix pipe idle.  The plain Gram route (rdyn_duo_gram.hip + Cholesky) runs
// at 2.8 ms but squares the condition number.  This file sits between the two -- preconditioned CholeskyQR whose result is accepted
// by the device only on MEASURED quality:
//
//   pass A   the Gram matrix of a ROW SUBSAMPLE: every S-th 16-sample tile, ~1 000 tiles whatever the batch size, through the
//            regressor -> Gram kernel of rdyn_duo_gram.hip (one tile per wave pair, ~30 us with its slab reduction).
//   precond  k_cholqr_precond: Cholesky of that matrix scaled to all rows, with nearly dependent pivots DEFERRED (a regressor is
//            structurally rank deficient; a tiny pivot used for elimination puts 1 / pivot into W, which amplifies the rounding of
//            A W) -> triangular T, W = T^-1 (carried along by the factorisation), in MFMA operand order; the growth factor gamma.
//   pass B   k_regressor_pgram: ALL rows.  The wave-pair design of rdyn_duo_gram.hip: the sweeper wave drops every finished link's
//            regressor rows into the pair's LDS tile; the consumer wave multiplies each 16-row group by W (v_mfma_f64_16x16x4_f64,
//            W in LDS in operand order) and accumulates the Gram of the PRODUCT straight from the result registers (the D layout of
//            the first MFMA is the A/B operand layout of the second: no transposition, no LDS round trip).  Q = A W is never stored.
//            Block-triangular zero band of a row group: preserved by the upper-triangular W, skipped in both stages.
//   factor   k_cholqr_factor: R2 = chol(G2), R = R2 T.  R'R = [A b]'[A b] for ANY invertible triangular T in exact arithmetic; in
//            floating point the error is u gamma (rounding of A W carried back by T) + u cond^2 of the column-equilibrated Q.  Both
//            are evaluated on ALL rows here (gamma on the column norms of R, rho = |Re^-1|_F / sqrt(k) of the equilibrated factor),
//            the deferred columns are decided here (residue -> exactly zero row of R; anything else is a pivot), and a round that
//            cannot vouch for its result raises a device flag:
//   round 1  the same four kernels with W from round 0's R (always queued, leave at once when the flag is clear);
//   stand-by the Householder folds of rdyn_tsqr.hip over all rows (always queued behind round 1, leave at once unless round 1 was not
//            accepted either, or a preconditioner was called off): the call as a whole is as robust as that route.
//
// fp64 throughout.  The multiplication by an explicit inverse (instead of a triangular solve) keeps every step a matrix product;
// its rounding is what gamma measures.  tools/cholqr_emulate.py replays the dense steps in numpy.
// k_cholqr_expand: factor of the reduced chain -> factor of a chain with fixed joints; k_cholqr_fold: the accumulate step.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <type_traits>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_gram_common.h"
#include "rdyn_duo_common.h"

#ifndef RDYN_CHOLQR_AHEAD
#define RDYN_CHOLQR_AHEAD 2  // rows of W operands requested ahead of their MFMAs when W is read from global memory (1 / 2 / 3: 6.47 / 6.33 / 6.36 ms at config 3)
#endif
#define DUO_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DUO_BARRIER() asm volatile("s_barrier" ::: "memory")

namespace
{

typedef const __attribute__((address_space(1))) double* GlobalD;

// pass B.  NPAIR wave pairs per workgroup (4, or 3 when four LDS tiles do not fit beside W: 7 joints).  Waves are dealt to the four
// SIMDs cyclically: with 8 waves pair p = (wave p, wave p + 4) shares SIMD p; with 6 waves the pairs are (0, 4), (1, 5) and (2, 3).
// NPAIR = 2 (7 joints + component columns: 21 accumulator tiles + 6 product tiles do not fit the 256 registers of a wave that shares
// its SIMD): four waves, every wave alone on its SIMD with up to 512 registers, pairs (0, 2) and (1, 3); fp64 MFMA and fp64 VALU
// exclude each other on a SIMD anyway, so two pairs on four SIMDs have the issue capacity of four pairs that
// share -- what is lost is the latency hiding of the second wave.
// Every chain joint is an input joint, in chain order (the reduced companion of a chain with fixed joints qualifies).
// WGLOBAL: W stays in global memory (30 KB at 7 joints: L1 / L2 resident) and the consumer loads its operands from there -- the LDS
// then holds four tiles again where W + four tiles exceed 160 KB (7 joints).
// XB = 1: the per-joint component columns of rdyn_identification_tsqr ride in the tile ([Y | C | tau_meas], one more 16-column block), as
// in rdyn_duo_gram.hip: a component column is stored as ONE 16-row group, that of its own joint.
// KIN != 0 (NPAIR = 4; KIN = the padding of the tile columns in doubles, 4 or 2): the one-lane-per-sample sweepers of rdyn_kin_sweepers.inc
// instead of the four row-pair sweepers -- wave 0 the link kinematics of the workgroup's 64 samples, waves 1-3 the rows (three each at
// most), waves 4-7 the consumers as before (same tiles bit for bit, same barriers + the sweepers' prologue).  The exchange area sits
// behind the four tiles.
template <int NJ, bool ALLREV, int NPAIR, bool WGLOBAL, int XB, int KIN = 0>
__global__ __launch_bounds__(128 * NPAIR) void k_regressor_pgram(const RdynLdsGramArgs fa, const double* __restrict__ Wg, const int* __restrict__ run_flag)
{
  constexpr bool DIRECT = true;
  constexpr int NB = (10 * NJ + 1 + 15) / 16 + XB, NT = NB * (NB + 1) / 2, P = 10 * NJ;
  // The consumer's column space is the natural order [links | tau_meas] SHIFTED RIGHT by SH: the padding of the last 16-column block
  // sits in front, where it joins the zero band of every row group (row group j is zero in the 10 j columns of the links upstream of
  // its joint): floor((SH + 10 j) / 16) whole blocks are skipped instead of floor(10 j / 16) -- 384 instead of 496 MFMAs per tile at
  // 7 joints (SH = 9), 264 instead of 288 at 6 (SH = 3) -- and W stays upper triangular (the descending order of rdyn_duo_gram.hip
  // buys the same blocks for a Gram, but a triangular factor fills a zero band on the right).  With component columns (K known at
  // run time only) SH = 0.
  constexpr int SH = XB > 0 ? 0 : 16 * NB - (P + 1);
  constexpr int WB = WGLOBAL ? 0 : NT * 2048;  // W in operand order: per (cb1 <= cb2) block four k-steps of 64 doubles
  if (run_flag && *run_flag == 0) return;  // second round not needed (uniform: every wave leaves)
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool sweeper = wave < NPAIR;
  const int pair = NPAIR == 4 ? (wave & 3) : NPAIR == 2 ? (wave & 1) : (wave < 3 ? wave : (wave == 3 ? 2 : wave - 4));
  if constexpr (!WGLOBAL)
  {
    double* const wl = (double*)lds_raw;
    for (int i = threadIdx.x; i < NT * 256; i += 128 * NPAIR) wl[i] = Wg[i];
    __syncthreads();
  }
  char* const tile = lds_raw + WB + (size_t)pair * fa.tile_bytes;  // shared by the pair
  const int n = fa.n_active;
  // tile_stride > 1: only every tile_stride-th 16-sample tile (the subsample pass); tile index t below counts the tiles that are swept
  const int64_t t_mul = fa.tile_stride > 1 ? fa.tile_stride : 1;
  const int64_t n_tiles = ((fa.n_samples + 15) / 16 + t_mul - 1) / t_mul;
  const int64_t t_step = (int64_t)gridDim.x * NPAIR;
  const int64_t t_first = (int64_t)blockIdx.x * NPAIR + pair;
  const int64_t trips_raw = (n_tiles - (int64_t)blockIdx.x * NPAIR + t_step - 1) / t_step;
  const int64_t trips = trips_raw > 0 ? trips_raw : 0;

#ifndef RDYN_PGRAM_X_IDLE
#define RDYN_PGRAM_X_IDLE(wave_) true
#endif
#ifdef RDYN_PGRAM_X_NOSWEEP  // timing experiment only (wrong numbers): the sweepers (those RDYN_PGRAM_X_IDLE names) keep the barriers and do nothing else
  if (sweeper && RDYN_PGRAM_X_IDLE(wave))
  {
    if (KIN) DUO_BARRIER();
    for (int64_t it = 0; it < trips; ++it)
      for (int f = 0; f <= NJ; ++f) DUO_BARRIER_LDS();
  }
  else
#endif
  if (sweeper && KIN)
  {
    // ================================================================ one lane per sample (doubles per sample and exchange buffer: 21 = R, the
    // joint offset, w, al, d; 12 at 7 joints: sin, 1 - cos, the displacement, w, al, d -- what the compact tiles leave of the LDS)
    constexpr int XV = NJ <= 6 ? 21 : 12;
    constexpr int KIN_SLOTS = 3;
    const int KIN_SW = wave;
    char* const kin_tiles = lds_raw + WB;
    double* const kin_xch_base = (double*)(lds_raw + WB + (size_t)4 * fa.tile_bytes);
#include "rdyn_kin_sweepers.inc"
  }
  else if (sweeper)
  {
    // ================================================================ sweeper: as in rdyn_duo_gram.hip (16 samples x 4 lanes)
    ChainPtr c = as_const(fa.chain);
    const int s_loc = lane >> 2, k = lane & 3;
    const int r0 = k, r1 = k + 4;
    RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
    const int fB = 4;
    double nqa = 0.0, ndqa = 0.0, nddqa = 0.0, nqb = 0.0, ndqb = 0.0, nddqb = 0.0, nb0 = 0.0, nb1 = 0.0;
    auto fetch = [&](int64_t tile_index) {
      int64_t sx = tile_index * t_mul * 16 + s_loc;
      if (sx >= fa.n_samples) sx = fa.n_samples - 1;
      const int64_t o = sx * fa.in_ss;
      if (fa.bcol)
      {
        if (r0 < n) nb0 = fa.bcol[o + in_oa];
        if (r1 < n) nb1 = fa.bcol[o + in_ob];
      }
      if (k < n)
      {
        nqa = fa.q[o + in_oa];
        ndqa = fa.dq[o + in_oa];
        nddqa = fa.ddq[o + in_oa];
      }
      if (k + 4 < n)
      {
        nqb = fa.q[o + in_ob];
        ndqb = fa.dq[o + in_ob];
        nddqb = fa.ddq[o + in_ob];
      }
    };
    if (t_first < n_tiles) fetch(t_first);
    for (int64_t it = 0; it < trips; ++it)
    {
      const int64_t tl = t_first + it * t_step;
      const bool valid = tl < n_tiles && tl * t_mul * 16 + s_loc < fa.n_samples;
      const int m0idx = valid ? r0 : -2, m1idx = valid ? r1 : -2;
      const double qa = nqa, dqa = ndqa, ddqa = nddqa, qb = nqb, dqb = ndqb, ddqb = nddqb;
      const double tb0 = valid ? nb0 : 0.0, tb1 = valid ? nb1 : 0.0;
      if (tl + t_step < n_tiles) fetch(tl + t_step);
      double sna, csa, snb, csb;
      rdyn_sincos(qa, &sna, &csa);
      rdyn_sincos(qb, &snb, &csb);
      const double oca = 1.0 - csa, ocb = 1.0 - csb;
      V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
      V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
      V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);
#pragma unroll
      for (int f = 0; f < NJ; ++f)
      {
#include "rdyn_duo_link_body.inc"
      }
      if constexpr (XB > 0)
      {
        // the consumer has read every row group of the previous tile by now (the last link's barrier is behind us)
#include "rdyn_duo_comp_cols.inc"
      }
      {
        char* const lb = tile + fa.lds_off_b + s_loc * 8;
        if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
        if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
      }
      DUO_BARRIER_LDS();  // the tile is complete
    }
  }
  else
  {
    // ================================================================ consumer: row group x W, then the Gram of the product
#ifdef RDYN_CHOLQR_MFMA_PRIO
    __builtin_amdgcn_s_setprio(RDYN_CHOLQR_MFMA_PRIO);  // A/B: the consumer is the bound of this kernel
#endif
    const int cl = lane & 15, g = lane >> 4;
    // stage 1 operand A: lane (cl, g) supplies X[sample cl][column 16 cb1 + 4 kk + g] of the row group.  Direct chains: link f's first
    // column sits at byte 640 f^2 + 960 f of the tile, its columns are 128 f + 160 bytes apart and hold row groups 0 .. f; the measured
    // torque (column P) holds every row group, columns beyond are padding.  Computed per operand (a handful of 32-bit instructions
    // under the MFMAs) instead of kept in 20 registers.
    // columns from P on ([C (K) | tau_meas | padding]; K = 0 without components): per-lane offset and row group of the (cb1, kk)
    // operands that can reach them, looked up once (the component table is a kernel argument indexed by a lane-dependent column)
    constexpr int X0 = (P + SH) / 4, NX = 4 * NB - X0;  // operand ids 4 cb1 + kk >= X0 touch (natural) columns >= P
    const int K = XB > 0 ? fa.n_comp_cols : 0;
    int xoff[NX], xrow[NX];  // xrow: the one row group stored, -1 = every row group (tau_meas), -2 = nothing (padding / a link column)
#pragma unroll
    for (int i = 0; i < NX; ++i)
    {
      const int col = 4 * (X0 + i) + g - SH;
      int off = 0, row = -2;
      if (col >= P && col < P + K)
      {
        row = fa.comp_col_row[col - P];
        off = fa.lds_off_c + (col - P) * fa.comp_stride - row * 128;
      }
      else if (col == P + K)
      {
        off = fa.lds_off_b;
        row = -1;
      }
      xoff[i] = off + cl * 8;
      xrow[i] = row;
    }
    auto a_operand = [&](int cb1, int kk, int j) -> double {
      const int id = 4 * cb1 + kk, col = 4 * id + g - SH;  // natural column; < 0: the padding in front
      double a = 0.0;
      if (id < X0 || (4 * id - SH < P && col < P))  // a link column (the second test: an operand that straddles P)
      {
        if (col < 0) return 0.0;
        const int f = (col * 205) >> 11;  // col / 10 for col < 1024
        constexpr int PAD = KIN ? KIN : 4;  // doubles of column padding (stride 128 f + 128 + 8 PAD bytes)
        const int off = f * (640 * f + 640 + 80 * PAD) + (col - 10 * f) * (128 * f + 128 + 8 * PAD);
        if (j <= f) a = *(const double*)(tile + off + cl * 8 + j * 128);
      }
      else
      {
        const int i = id - X0;
        if (xrow[i] == -1 || xrow[i] == j) a = *(const double*)(tile + xoff[i] + j * 128);
      }
      return a;
    };
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const char* const wl = lds_raw + lane * 8;   // W in LDS
    // W in global memory: uniform base (laundered per row group: hoisted out of the tile loop, the ~250 operand addresses of a tile
    // become 64-bit VGPR pairs and spill) + lane index
    // (explicitly a GLOBAL pointer: the laundering below hides the kernel argument it came from, and a flat load -- which may address
    // the LDS -- counts on both wait counters and returns out of order with LDS reads: every wait then becomes vmcnt(0) lgkmcnt(0),
    // the rows requested ahead included)
    GlobalD wgp = (GlobalD)Wg;
#ifdef RDYN_PGRAM_X_NOW  // timing experiment only (wrong numbers): no loads of W
    auto wglob = [&](int blk_kk) -> double { return (double)(blk_kk + lane); };
#else
    auto wglob = [&](int blk_kk) -> double { return (wgp + blk_kk * 64)[lane]; };
#endif
    d4 D[NB];
    // Q(rows of group j, :) = X(rows of group j, :) W;  D[cb2] register r of lane (cl, g) = Q[sample g + 4 r][16 cb2 + cl]
    auto stage1 = [&](int j, int band) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) D[cb] = (d4){0.0, 0.0, 0.0, 0.0};
      // rows (cb1, kk) of W in the order they are used, from the first block row of the band, RDYN_CHOLQR_AHEAD rows in flight: the
      // operands of row r + AHEAD -- the tile's own (an LDS read: ~130 cycles before a dependent MFMA may issue, 84 rows per tile at 7
      // joints: a third of the consumer's time when it was read in front of its MFMAs) and the row of W (LDS, or global memory: L1 / L2
      // resident) -- are requested before the MFMAs of row r are issued, and no further ahead (compiler barrier): left alone the
      // scheduler hoists every load of the group and spills the accumulators
      constexpr int AH = RDYN_CHOLQR_AHEAD;
      double ring[AH + 1][NB], aring[AH + 1];
      if constexpr (WGLOBAL) asm volatile("" : "+s"(wgp));
      auto load_row = [&](int r, double (&dst)[NB], double& adst) {  // r = flat row index from the band's first row
        const int c1 = band + (r >> 2), k4 = r & 3;
        adst = c1 < NB ? a_operand(c1, k4, j) : 0.0;
#pragma unroll
        for (int cb2 = 0; cb2 < NB; ++cb2)
        {
          dst[cb2] = 0.0;
          if (c1 < NB && cb2 >= c1)
          {
            if constexpr (WGLOBAL) dst[cb2] = wglob((cb2 * (cb2 + 1) / 2 + c1) * 4 + k4);
            else dst[cb2] = *(const double*)(wl + ((cb2 * (cb2 + 1) / 2 + c1) * 4 + k4) * 512);
          }
        }
      };
#pragma unroll
      for (int r = 0; r < AH; ++r) load_row(r, ring[r], aring[r]);
#pragma unroll
      for (int cb1 = 0; cb1 < NB; ++cb1)
      {
        if (cb1 < band) continue;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
        {
          const int r = (cb1 - band) * 4 + kk;
          load_row(r + AH, ring[(r + AH) % (AH + 1)], aring[(r + AH) % (AH + 1)]);
          asm volatile("" ::: "memory");
          const double a = aring[r % (AH + 1)];
#pragma unroll
          for (int cb2 = cb1; cb2 < NB; ++cb2) D[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ring[r % (AH + 1)][cb2], D[cb2], 0, 0, 0);
        }
      }
    };
    // acc(rb, cb) += Q(:, block rb)' Q(:, block cb): register r of D is k-step r (which rows share a k-step is irrelevant to the sum)
    auto stage2 = [&](int band) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
        int ti = 0;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
          for (int rb = 0; rb <= cb; ++rb)
          {
            if (rb >= band) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[rb][t], D[cb][t], acc[ti], 0, 0, 0);
            ++ti;
          }
      }
    };
    if (KIN) DUO_BARRIER();  // the prologue of the one-lane-per-sample sweepers (link 0 of the first tile is published)
    for (int64_t it = 0; it <= trips; ++it)
    {
#ifdef RDYN_PGRAM_X_NOMFMA  // timing experiment only (wrong numbers): the consumers keep the barriers and do nothing else
      const bool have = it > trips + 1;
#else
      const bool have = it > 0;  // the tile in LDS is complete (nothing to consume while the first tile is being swept)
#endif
#pragma unroll
      for (int f = 0; f < NJ; ++f)
      {
        if (have) stage1(f, (SH + 10 * f) >> 4);
        // my reads of row group f have returned -> the sweeper may overwrite link f's columns (no later group reads them)
        if (it < trips) DUO_BARRIER_LDS();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (have) stage2((SH + 10 * f) >> 4);
      }
      if (it < trips) DUO_BARRIER_LDS();  // end of the sweeper's tile
    }
    // ---- block reduction of the consumers (fixed order); the reduction area overlays W (every consumer is done with it)
    DUO_BARRIER_LDS();
    double* red = (double*)lds_raw;
    for (int w = 0; w < NPAIR; ++w)
    {
      if (pair == w)
      {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
          {
            const int idx = t * 256 + ((g + 4 * r) * 16 + cl);
            red[idx] = (w == 0) ? acc[t][r] : red[idx] + acc[t][r];
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (sweeper)
    for (int w = 0; w < NPAIR + 1; ++w) __builtin_amdgcn_s_barrier();  // the sweepers take part in the barriers of the reduction
  asm volatile("" ::: "memory");
  {
    const double* red = (const double*)lds_raw;
    double* slab = fa.slabs + (int64_t)blockIdx.x * (NT * 256);
    for (int i = threadIdx.x; i < NT * 256; i += 128 * NPAIR) slab[i] = red[i];
  }
}

// pass B for a MATERIALISED matrix (rdyn_tsqr): 16-row groups of the column-major rows x n_cols device matrix [A | b] (b may be null:
// n1 = n_cols + (b != null) columns, n1 <= 16 NB) times W, then the Gram of the product -- the consumer of k_regressor_pgram fed from
// memory instead of a sweeper's LDS tile.  Operand A of the first MFMA: lane (cl, g) supplies X[row r0 + cl][column 16 cb1 + 4 kk + g]:
// per (cb1, kk) the wave reads 16 consecutive rows (128 contiguous bytes) of four columns; the next group's 4 NB operands are
// requested behind this group's MFMAs.  No zero band (nothing is known about the matrix), natural column order.  Four waves per
// workgroup, one per SIMD (NB (NB + 1) / 2 accumulator tiles + NB product tiles: up to 432 registers at NB = 6), W in LDS.
template <int NB>
__global__ __launch_bounds__(256) void k_pgram_rows(const double* __restrict__ A, const double* __restrict__ bvec, int64_t rows, int64_t lda, int n_cols,
                                                    const double* __restrict__ Wg, double* __restrict__ slabs, const int* __restrict__ run_flag)
{
  constexpr int NT = NB * (NB + 1) / 2;
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = lane & 15, g = lane >> 4;
  {
    double* const wl = (double*)lds_raw;
    for (int i = threadIdx.x; i < NT * 256; i += 256) wl[i] = Wg[i];
    __syncthreads();
  }
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  const char* const wl = lds_raw + lane * 8;
  const int64_t gstride = (int64_t)gridDim.x * 4 * 16;
  int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  double cur[4 * NB], nxt[4 * NB];
  // operand id (= 4 cb1 + kk) of lane (cl, g): column 4 id + g -- one per-lane base (column g) + a wave-uniform column step, so that
  // the 4 NB addresses cost no registers; the right-hand side (column n_cols) sits in one operand id only
  const double* const base = A + (int64_t)g * lda;
  const int id_b = bvec ? n_cols >> 2 : -1, g_b = n_cols & 3;
  auto load = [&](int64_t rbase, double (&v)[4 * NB]) {
    const int64_t r = rbase + cl;
    const bool in = r < rows;
    const double* pcol = base + r;  // walks the columns g, g + 4, ... (one address register pair for the 4 NB loads)
#pragma unroll
    for (int id = 0; id < 4 * NB; ++id)
    {
      double x = 0.0;
      if (in && 4 * id + g < n_cols) x = *pcol;
      if (id == id_b && in && g == g_b) x = bvec[r];
      v[id] = x;
      pcol += 4 * lda;
      asm volatile("" : "+v"(pcol));
    }
  };
  if (r0 < rows) load(r0, cur);
  while (r0 < rows)
  {
    const int64_t rn = r0 + gstride;
    // seven column blocks: 28 accumulator tiles + this group's operands + the products + the ring leave no room for the next group's
    // operands (268 B of scratch in the loop, 3x the time per row): they are requested behind stage 1, under the 112 MFMAs of stage 2
    constexpr bool LATE = NB >= 7;
    if (!LATE && rn < rows) load(rn, nxt);
    d4 D[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) D[cb] = (d4){0.0, 0.0, 0.0, 0.0};
    {
      // rows (cb1, kk) of W from LDS, two rows ahead of their MFMAs and no further (compiler barrier): left alone the scheduler hoists
      // every operand of the group and spills the accumulators at six column blocks
      constexpr int AH = LATE ? 1 : 2;
      double ring[AH + 1][NB];
      auto load_row = [&](int r, double (&dst)[NB]) {
        const int c1 = r >> 2, k4 = r & 3;
#pragma unroll
        for (int cb2 = 0; cb2 < NB; ++cb2)
        {
          dst[cb2] = 0.0;
          if (c1 < NB && cb2 >= c1) dst[cb2] = *(const double*)(wl + ((cb2 * (cb2 + 1) / 2 + c1) * 4 + k4) * 512);
        }
      };
#pragma unroll
      for (int r = 0; r < AH; ++r) load_row(r, ring[r]);
#pragma unroll
      for (int cb1 = 0; cb1 < NB; ++cb1)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
        {
          const int r = cb1 * 4 + kk;
          load_row(r + AH, ring[(r + AH) % (AH + 1)]);
          asm volatile("" ::: "memory");
          const double a = cur[r];
#pragma unroll
          for (int cb2 = cb1; cb2 < NB; ++cb2) D[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ring[r % (AH + 1)][cb2], D[cb2], 0, 0, 0);
        }
    }
    if (LATE)
    {
      asm volatile("" ::: "memory");
      if (rn < rows) load(rn, nxt);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
    {
      int ti = 0;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int rb = 0; rb <= cb; ++rb)
        {
          acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[rb][t], D[cb][t], acc[ti], 0, 0, 0);
          ++ti;
        }
    }
#pragma unroll
    for (int id = 0; id < 4 * NB; ++id) cur[id] = nxt[id];
    r0 = rn;
  }
  __syncthreads();  // everybody is done with W: the reduction area overlays it
  gram_block_reduce_to_slab<NT>(acc, (double*)lds_raw, wave, cl, g, slabs + (int64_t)blockIdx.x * (NT * 256), false);
}

// ---------------------------------------------------------------- the small dense steps (one workgroup each, n1 <= 81)
constexpr int kMaxN1 = 112;    // widest factor of the dense kernels
constexpr int kMaxN1Lds = 96;  // ... with their two n1 x n1 squares in LDS (147 KB); wider: the squares live in the workspace (WIDE)
constexpr int kMaxFoldN1 = 136;  // k_cholqr_fold: two packed triangles in LDS (136 * 137 * 8 = 149 KB)
#ifndef RDYN_CHOLQR_DENSE_THREADS
#define RDYN_CHOLQR_DENSE_THREADS 1024
#endif
#ifdef RDYN_CHOLQR_STAMPS  // timing experiments: phase stamps (100 MHz wall clock) behind the diagnostics of the factor kernel
#define STAMP(i) do { if (threadIdx.x == 0 && rho_out) rho_out[4 + (i)] = (double)wall_clock64(); } while (0)
#define STAMP_P(i) do { if (threadIdx.x == 0 && gamma_out) gamma_out[14 + (i)] = (double)wall_clock64(); } while (0)  // (round 0: flag doubles 70 ..)
#else
#define STAMP(i) do { } while (0)
#define STAMP_P(i) do { } while (0)
#endif
#ifdef RDYN_CHOLQR_CHOL_LDS  // A/B builds: the factorisation that keeps its matrices in LDS (rounds 3, 4)
#define RDYN_CHOL_WITH_INVERSE chol_with_inverse_lds
#else
#define RDYN_CHOL_WITH_INVERSE chol_with_inverse_regs
#endif
constexpr int NTD = RDYN_CHOLQR_DENSE_THREADS;  // threads of the single-workgroup dense kernels (256 / 512 / 1024: 1.45 / 1.41 / 1.40 ms for the whole call at config-2 size; >= 128)

// Householder QR of the m x nc matrix B (column-major, leading dimension m) in LDS, in place, by the NTD threads of the workgroup:
// R in the upper triangle; what is left BELOW the diagonal is the reflectors, not zeros -- callers read the upper triangle only.
// One barrier per column step: every wave forms the column's norm by itself (the same sum in the same order: the waves agree to the
// bit), the reflector is read in place, the new diagonal is written behind the step's barrier.
// n_steps: columns that are reduced (the remaining ones only have the reflections applied to them); < 0: min(nc, m).
// row_end (LDS, may be null): per reduced column k, one past the last row that can be non-zero below the diagonal (a triangular factor
// with columns removed: the reflector of step k ends at the original index of its column).
__device__ __forceinline__ void small_qr_lds(double* B, int m, int nc, int tid, int n_steps = -1, const int* row_end = nullptr)
{
  const int steps = n_steps >= 0 ? n_steps : (nc < m ? nc : m);
  const int lane = tid & 63;
  for (int k = 0; k < steps; ++k)
  {
    const int rend = row_end ? (row_end[k] < m ? row_end[k] : m) : m;
    double sigma = 0.0;
    for (int r = k + 1 + lane; r < rend; r += 64) sigma = fma(B[k * m + r], B[k * m + r], sigma);
    for (int o = 32; o > 0; o >>= 1) sigma += __shfl_xor(sigma, o);
    const double alpha = B[k * m + k];
    double scale = 0.0, v0 = 0.0, beta = alpha;
    if (sigma > 1e-280)  // (else: nothing below the diagonal, no reflection)
    {
      const double norm = sqrt(fma(alpha, alpha, sigma));
      beta = alpha > 0.0 ? -norm : norm;
      v0 = alpha - beta;
      scale = 2.0 / fma(v0, v0, sigma);
      // columns j > k: four threads per column share the rows
      const int ncol = nc - k - 1;
      for (int e = tid; e < (ncol * 4 + NTD - 1) / NTD * NTD; e += NTD)
      {
        const bool on = e < ncol * 4;
        const int j = on ? k + 1 + (e >> 2) : k, q = e & 3;
        double d = (on && q == 0) ? v0 * B[j * m + k] : 0.0;
        if (on)
          for (int r = k + 1 + q; r < rend; r += 4) d = fma(B[k * m + r], B[j * m + r], d);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        const double f = scale * d;
        if (on)
        {
          if (q == 0) B[j * m + k] = fma(-f, v0, B[j * m + k]);
          for (int r = k + 1 + q; r < rend; r += 4) B[j * m + r] = fma(-f, B[k * m + r], B[j * m + r]);
        }
      }
    }
    __syncthreads();
    if (tid == 0) B[k * m + k] = beta;
  }
  __syncthreads();
}

// X = U^-1 for an upper-triangular U (n x n, column-major, leading dimension n, in LDS; rdiag = 1 / diag(U), in LDS); X in LDS, its
// strict lower triangle untouched.  Four lanes per column: lane q of a column owns the entries x(k), k = q mod 4 (it writes and later
// reads them itself: no exchange through LDS inside the wave), the inner sums meet by two shuffles.  Long columns first.
// fro2 (LDS, may be null): per column the sum of squares of its entries of X.  Callers put barriers around the call.
__device__ __forceinline__ void tri_inverse_lds(const double* U, const double* rdiag, double* X, int n, int tid, double* fro2)
{
  const int q = tid & 3, lane = tid & 63;
  for (int c0 = 0; c0 < n; c0 += NTD / 4)
  {
    const int c = n - 1 - c0 - (tid >> 2);  // this lane's column (< 0: none)
    const int c_top = n - 1 - c0 - ((tid - lane) >> 2);  // the longest column of the wave: its trip count for everyone (uniform shuffles)
    double own2 = 0.0;
    if (c >= 0 && (c & 3) == q)
    {
      const double x = rdiag[c];
      X[c * n + c] = x;
      own2 = x * x;
    }
    for (int i = c_top - 1; i >= 0; --i)
    {
      const bool on = c >= 0 && i < c;
      double s = 0.0;
      if (on)
      {
        int k = i + 1 + ((q - (i + 1)) & 3);  // the first k > i with k = q mod 4
        for (; k <= c; k += 4) s = fma(U[k * n + i], X[c * n + k], s);
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (on && (i & 3) == q)
      {
        const double x = -s * rdiag[i];
        X[c * n + i] = x;
        own2 = fma(x, x, own2);
      }
    }
    own2 += __shfl_xor(own2, 1);
    own2 += __shfl_xor(own2, 2);
    if (fro2 && c >= 0 && q == 0) fro2[c] = own2;
  }
}

// Cholesky-type factorisation of the symmetric matrix in the upper triangle of M (n x n, column-major, LDS) TOGETHER with the inverse
// of the factor: an identity rides along in E (n x n, LDS) and receives the same row operations (forward substitution), so the
// dependent chain of the factorisation pays for both.  One barrier per eliminating pivot: everybody evaluates the (uniform) decision
// decide(k, d) -> eliminate or not, diagf(k, d, elim) -> the diagonal; the trailing updates work from the UNSCALED row k while the
// scaled rows are parked where nothing reads them during the loop:
//   row k of the factor F (F'F = M on the eliminating pivots)  -> column k of the strict LOWER triangle of M, diagonal in fdiag[k]
//   row k of F^-T = column k of F^-1                           -> column k of the strict UPPER triangle of E, diagonal in xdiag[k]
// A pivot that does not eliminate (elim = 0) leaves row k of F as diag e_k' (diag = 0: the row is left out altogether, X too).
// In: E = identity (strict upper triangle zero), strict lower triangle of M zero.  Callers put a barrier in front.
// (Two pivots per barrier -- a rank-2 update with the second pivot's row formed on the fly -- was measured: 51 us instead of 43 us per
// 61 pivots.  A step is a chain of three or four LDS round trips, a square root and a division; the barrier is the small part.)
struct PivotAct
{
  int elim;
  double diag;
};
template <int MAXNB = 6, class Decide, class Diag, class Idle>
__device__ __forceinline__ void chol_with_inverse_lds(double* M, double* E, int n, int tid, double* fdiag, double* xdiag, Decide decide, Diag diagf, Idle idle_work)
{
  const int tx = tid & 31, ty = tid >> 5;
  if (tid >= 384) idle_work();
  for (int k = 0; k < n; ++k)
  {
    const double d = M[k * n + k];
    PivotAct act;
    act.elim = decide(k, d) ? 1 : 0;
    act.diag = diagf(k, d, act.elim != 0);
    if (tid == 0)
    {
      fdiag[k] = act.diag;
      xdiag[k] = act.diag > 0.0 ? 1.0 / act.diag : 0.0;
    }
    if (!act.elim)
    {
      // row k of E has seen the eliminations of the pivots before it and is final: X(k, :) = E(k, :) / diag
      if (act.diag > 0.0)
      {
        const double inv = 1.0 / act.diag;
        for (int c = tid; c < k; c += NTD) E[k * n + c] = E[c * n + k] * inv;
      }
      continue;
    }
    const double inv_d = 1.0 / d, inv_p = act.diag * inv_d;  // 1 / sqrt(d) = sqrt(d) / d
    // 32 x (NTD / 32) threads over (i, j) / (i, c): no integer divisions in the dependent chain
    for (int j = k + 1 + ty; j < n; j += NTD / 32)  // M(i, j) -= M(k, i) M(k, j) / d, k < i <= j
    {
      const double f = M[j * n + k] * inv_d;
      for (int i = k + 1 + tx; i <= j; i += 32) M[j * n + i] = fma(-M[i * n + k], f, M[j * n + i]);
    }
    for (int c = ty; c <= k; c += NTD / 32)  // E(i, c) -= M(k, i) / d E(k, c), i > k, c <= k
    {
      const double f = E[c * n + k] * inv_d;
      for (int i = k + 1 + tx; i < n; i += 32) E[c * n + i] = fma(-M[i * n + k], f, E[c * n + i]);
    }
    for (int j = k + 1 + tid; j < n; j += NTD) M[k * n + j] = M[j * n + k] * inv_p;
    for (int c = tid; c < k; c += NTD) E[k * n + c] = E[c * n + k] * inv_p;
    __syncthreads();
  }
  __syncthreads();
}

// The same factorisation with the working matrices in REGISTERS (round 4).  The LDS version spends 0.7 us per pivot at ANY thread count:
// every one of its 16 waves repeats the pivot's square root and division and streams its share of ~ (n - k) n elements through LDS, a
// dependent LDS round trip per loop level.
//   waves 0-3   M's upper triangle and E's lower triangle as one n x n square on a 16 x 16 thread grid (thread (ty, tx) owns rows
//               16 a + ty, columns 16 b + tx).  The owners of row k publish it once per step -- rM(x) = M(k, x) for x > k, else 0;
//               rE(x) = E(k, x) for x < k, 1 at x = k, else 0;  d = M(k, k) -- and with those zeros the trailing update needs no masks:
//               f_i = rM(i) / d vanishes for i <= k,  M(i, x) -= f_i rM(x) (blocks b >= a),  E(i, x) -= f_i rE(x) (blocks a >= b);
//               what that leaves in the halves of the diagonal blocks that belong to the other triangle is never published.  The block
//               row AK = k / 16 of the pivot is a compile-time constant of the step's code (six copies): which blocks are live, which
//               registers hold row k + 1 -- straight-line code, ALL the step's LDS reads in one round trip, one barrier per step, two row
//               buffers, one division and no square root in the dependent chain.
//   waves 4, 5  park the published rows UNSCALED where the LDS version parks the scaled ones.
//   the rest    keep the barrier count.
// Afterwards, in parallel: the diagonals (square roots), the row scales, one pass over the parked rows.  decide(k, d) -> eliminate or
// not (uniform; its side effects are thread 0's), diagf(k, d, elim) -> the diagonal of row k of the factor (pure); idle_work(): run by
// the waves that only keep the barrier count (threads 384 ..) before they do -- their barriers carry no fence, so global loads issued
// there stay in flight through the whole factorisation.
constexpr int kCholRS = 112;  // row buffer: rM | rE | d
template <int NB, class Decide>
__device__ __forceinline__ void chol_compute_waves(const double* M, const double* E, int n, int tid, double* fdiag, double* xdiag, double (*s_row)[2 * kCholRS + 2],
                                                   Decide decide)
{
  constexpr int RS = kCholRS;
  const int tx = tid & 15, ty = tid >> 4;
  double Mr[NB][NB], Er[NB][NB];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
    {
      const int i = 16 * a + ty, x = 16 * b + tx;
      const bool in = i < n && x < n;
      if (b >= a) Mr[a][b] = in ? M[x * n + i] : 0.0;
      if (a >= b) Er[a][b] = in ? E[x * n + i] : 0.0;
    }
  auto publish = [&](auto akc, int k) {
    constexpr int AK = decltype(akc)::value;  // the block row of row k
    if constexpr (AK < NB)
    {
      if (k >= n || ty != (k & 15)) return;
      double* const rb = s_row[k & 1];
#pragma unroll
      for (int b = 0; b < NB; ++b)
      {
        const int x = 16 * b + tx;
        if (b >= AK) rb[x] = x > k ? Mr[AK][b] : 0.0;
        if (b <= AK) rb[RS + x] = x < k ? Er[AK][b] : (x == k ? 1.0 : 0.0);
      }
      if (tx == (k & 15)) rb[2 * RS] = Mr[AK][AK];
    }
  };
  auto block_row = [&](auto akc) {
    constexpr int AK = decltype(akc)::value;
    for (int k = 16 * AK; k < 16 * AK + 16 && k < n; ++k)
    {
      __syncthreads();
      const double* const rb = s_row[k & 1];
      const double d = rb[2 * RS];
      double rm[NB], re[NB], fr[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
      {
        if (b >= AK) rm[b] = rb[16 * b + tx];
        if (b >= AK) fr[b] = rb[16 * b + ty];
        if (b <= AK) re[b] = rb[RS + 16 * b + tx];
      }
      const bool elim = decide(k, d);
      if (tid == 0)
      {
        fdiag[k] = d;  // (the diagonals and scales follow after the loop)
        xdiag[k] = elim ? 1.0 : 0.0;
      }
      // block row AK first: it holds row k + 1 (unless that is the first row of the next block row), whose owners publish it before
      // the other block rows are updated -- the LDS writes complete behind that arithmetic
      const bool next_in_block = k + 1 < 16 * AK + 16;
      double inv_d = 0.0;
      auto update_rows = [&](auto a0c, auto a1c) {
#pragma unroll
        for (int a = decltype(a0c)::value; a < decltype(a1c)::value; ++a)
        {
          const double fi = fr[a] * inv_d;
#pragma unroll
          for (int b = 0; b < NB; ++b)
          {
            if (b >= a) Mr[a][b] = fma(-fi, rm[b], Mr[a][b]);
            if (b <= AK) Er[a][b] = fma(-fi, re[b], Er[a][b]);
          }
        }
      };
      if (elim)
      {
        // 1 / d by v_rcp_f64 and two Newton steps (<= 1 ulp; the IEEE division's scaling and fix-up are another ~100 cycles of the chain)
        inv_d = __builtin_amdgcn_rcp(d);
        inv_d = fma(fma(-d, inv_d, 1.0), inv_d, inv_d);
        inv_d = fma(fma(-d, inv_d, 1.0), inv_d, inv_d);
        update_rows(std::integral_constant<int, AK>(), std::integral_constant<int, AK + 1>());
      }
      // (a pivot that does not eliminate leaves row k + 1 as it is)
      if (next_in_block) publish(std::integral_constant<int, AK>(), k + 1);
      if (elim) update_rows(std::integral_constant<int, AK + 1>(), std::integral_constant<int, NB>());
      if (!next_in_block) publish(std::integral_constant<int, AK + 1>(), k + 1);
    }
  };
  publish(std::integral_constant<int, 0>(), 0);
  block_row(std::integral_constant<int, 0>());
  if constexpr (NB > 1) block_row(std::integral_constant<int, 1>());
  if constexpr (NB > 2) block_row(std::integral_constant<int, 2>());
  if constexpr (NB > 3) block_row(std::integral_constant<int, 3>());
  if constexpr (NB > 4) block_row(std::integral_constant<int, 4>());
  if constexpr (NB > 5) block_row(std::integral_constant<int, 5>());
  if constexpr (NB > 6) block_row(std::integral_constant<int, 6>());
}
// MAXNB: the widest instantiation compiled in (6: n <= 96; 7: n <= 112 -- 98 working values per thread beside the row buffers: the kernels of
// 1 024 threads spill some of them, the WIDE kernels only).
template <int MAXNB = 6, class Decide, class Diag, class Idle>
__device__ __forceinline__ void chol_with_inverse_regs(double* M, double* E, int n, int tid, double* fdiag, double* xdiag, Decide decide, Diag diagf, Idle idle_work)
{
  static_assert(NTD >= 384 && NTD % 128 == 0 && kMaxN1Lds <= 96 && kMaxN1 <= 112 && MAXNB >= 6 && MAXNB <= 7, "four computing waves, two parking waves; six (seven) 16-wide blocks per side");
  constexpr int RS = kCholRS;
  __shared__ double s_row[2][2 * RS + 2], s_scm[kMaxN1], s_sce[kMaxN1];
  if (tid < 256)
  {
    if (n <= 64) chol_compute_waves<4>(M, E, n, tid, fdiag, xdiag, s_row, decide);
    else if (n <= 80) chol_compute_waves<5>(M, E, n, tid, fdiag, xdiag, s_row, decide);
    else if (n <= 96 || MAXNB < 7) chol_compute_waves<6>(M, E, n, tid, fdiag, xdiag, s_row, decide);
    else if constexpr (MAXNB >= 7) chol_compute_waves<7>(M, E, n, tid, fdiag, xdiag, s_row, decide);
  }
  else if (tid < 384)
  {
    const int l = tid & 63;
    for (int k = 0; k < n; ++k)
    {
      __syncthreads();
      const double* const rb = s_row[k & 1];
      if (tid < 320)
        for (int j = k + 1 + l; j < n; j += 64) M[k * n + j] = rb[j];
      else
        for (int c = l; c < k; c += 64) E[k * n + c] = rb[RS + c];
    }
  }
  else
  {
    idle_work();
    for (int k = 0; k < n; ++k) __builtin_amdgcn_s_barrier();  // (nothing of theirs to publish or to see: no fence)
  }
  __syncthreads();
  if (tid < n)
  {
    const double d = fdiag[tid];
    const bool elim = xdiag[tid] != 0.0;
    const double dg = diagf(tid, d, elim), inv = dg > 0.0 ? 1.0 / dg : 0.0;
    fdiag[tid] = dg;
    xdiag[tid] = inv;
    s_scm[tid] = elim ? dg / d : 0.0;   // 1 / sqrt(d) = sqrt(d) / d;  no elimination: row k of F is diag e_k'
    s_sce[tid] = elim ? dg / d : inv;  // no elimination: X(k, :) = E(k, :) / diag (left out altogether at diag = 0)
  }
  __syncthreads();
  {
    const int i = tid & 127;
    if (i < n)
      for (int k = tid >> 7; k < n; k += NTD / 128)
      {
        if (i > k) M[k * n + i] = s_scm[k] != 0.0 ? M[k * n + i] * s_scm[k] : 0.0;
        else if (i < k) E[k * n + i] = s_sce[k] != 0.0 ? E[k * n + i] * s_sce[k] : 0.0;
      }
  }
  __syncthreads();
}

// The preconditioner of a round.  In: an upper-triangular factor R1 (n1 x n1): round 0 the Householder factor of the row subsample,
// round 1 the factor round 0 produced.  W = T^-1 for a triangular T that makes Q = A W well conditioned -- ANY invertible upper-
// triangular T gives R = chol((A W)'(A W)) T with R'R = A'A in exact arithmetic.  In floating point Q = A W carries a rounding error
// u |A| |W| per row, which T multiplies back: column j of A is reproduced with a relative error of about
//     u gamma_j,   gamma_j = sum_l g_l |T(l, j)| / |a_j|,   g_l = sum_i |a_i| |W(i, l)|
// and a pivot T(k, k) that is tiny against its own column while later columns have components in its row puts 1 / T(k, k) into gamma.
// A regressor always has such pivots (it is structurally rank deficient: the pivot of a dependent column is rounding residue), and a
// subsample can have more of them than the batch.  So:
//   deferred set Z   column k is NOT used to eliminate later columns when its pivot is below 1e-13 x the largest column norm to its
//                    left (rounding residue of a dependent column: relative to what it depends on, not to itself) or below 1e-5 x its
//                    own norm (the last column -- the measured torque, nothing to its right -- excepted)
//   T                the factor re-triangularised WITHOUT the deferred columns (Householder QR of R1(:, not Z), in LDS), embedded back
//                    at the positions of the kept columns.  A deferred column k rides through the same reflections: its coefficients on
//                    the kept directions to its LEFT go into T(:, k), its diagonal is its own pivot (or the 1e-13 floor).  Row k of T
//                    holds the diagonal only, so W has no large rows; the large column W(:, k) touches nothing but column k itself.
//                    What the later columns keep of direction k is left to the Cholesky factorisation of Q'Q over ALL rows.
//   gamma            max_j gamma_j is evaluated on the finished T, W.  Above 1e4 (round 1; round 0, which may still serve as the
//                    preconditioner of round 1: 1e10) the round is not run: flags say so and the Householder factorisation of all
//                    rows (the stand-by call of rdyn_api.cpp) takes over.
// flags: [0] run round 1, [1] run the stand-by, [2] run round 0 (written here in round 0).  zmask <- Z (for the factor kernel).
// W is written in the MFMA operand order of k_regressor_pgram.
constexpr double kCholqrGammaMax = 1e4;
// WIDE (97 .. 112 columns): the two squares do not fit the LDS -- they live in `wide_sq` (2 n1^2 doubles of the workspace: L1 / L2
// resident; one workgroup, so its barriers order the accesses).  The factorisation itself works in registers as before (the squares are
// only where it starts from and where it parks its rows); the element-wise passes around it run at global-memory latency.
template <bool WIDE>
__global__ __launch_bounds__(NTD) void k_cholqr_precond(const double* __restrict__ R1, const double* __restrict__ Gs, const double* __restrict__ cs,
                                                        const double* __restrict__ bbs, int n1, int col_shift, int nb_w, double row_scale,
                                                        double* __restrict__ Tout, double* __restrict__ W, double* __restrict__ Vout, int* __restrict__ zmask,
                                                        int* __restrict__ flags, int round, const int* __restrict__ run_flag,
                                                        double* __restrict__ gamma_out, double* __restrict__ wide_sq)
{
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* const A0 = WIDE ? wide_sq : sh;  // [n1][n1] column-major: R1, then T
  double* const B = A0 + n1 * n1;          // [nc][n1]: the kept columns (or the Gram matrix being factorised), then V = T^-1
  __shared__ double s_part[kMaxN1], s_norm[kMaxN1], s_lift[kMaxN1], s_g[kMaxN1];
  __shared__ int s_z[kMaxN1], s_cmap[kMaxN1], s_rend[kMaxN1], s_nc;
  const int tid = threadIdx.x;
  STAMP_P(0);
  if (Gs)
  {
    // ---- from the Gram matrix [G c; c' bb] of the subsample: Cholesky in which a deferred pivot eliminates nothing.  Row k of T is
    // M(k, :) / pivot (the coefficients of the later columns, deferred ones included, on direction k); a deferred row holds its lift.
    // One barrier per pivot: everybody evaluates the (uniform) pivot decision, the scaled row goes to T, the trailing update works
    // from the unscaled one.
    const int P = n1 - 1;
    const double sc2 = row_scale * row_scale;  // the Gram matrix of all rows is row_scale^2 x that of the subsample
    for (int i = tid; i < n1 * n1; i += NTD)
    {
      const int r = i % n1, c = i / n1;
      double v;
      if (r < P && c < P) v = Gs[(int64_t)c * P + r];
      else if (r == P && c == P) v = bbs[0];
      else v = cs[r < P ? r : c];
      B[i] = r <= c ? sc2 * v : 0.0;
      A0[i] = r == c ? 1.0 : 0.0;  // the identity that becomes T^-1
    }
    __syncthreads();
    if (tid < n1) s_norm[tid] = sqrt(fmax(B[tid * n1 + tid], 0.0));
    __syncthreads();
    if (tid < n1)
    {
      double mx = 0.0;
      for (int c = 0; c <= tid; ++c) mx = fmax(mx, s_norm[c]);
      s_lift[tid] = mx > 0.0 ? 1e-13 * mx : 1.0;
    }
    __syncthreads();
    STAMP_P(1);
    auto chol = [&](auto... args) {
      if constexpr (WIDE) RDYN_CHOL_WITH_INVERSE<7>(args...);
      else RDYN_CHOL_WITH_INVERSE<6>(args...);
    };
    chol(B, A0, n1, tid, s_part, s_g, [&](int k, double d) {
      // the squared sine of the angle to the columns on the left: a Gram matrix resolves it down to ~1e-14; below 1e-10 (sine 1e-5, the
      // own-norm rule of the other branch) or below the residue floor the column is deferred.  The last column eliminates nothing.
      const double g0 = s_norm[k] * s_norm[k];  // d / g0 = the squared sine
      const bool defer = !(d >= (k + 1 < n1 ? 1e-10 : 1e-14) * g0) || !(g0 > 0.0) || !(d >= s_lift[k] * s_lift[k]);
      if (tid == 0) s_z[k] = defer ? 1 : 0;
      return !defer;
    }, [&](int k, double d, bool elim) { return elim ? sqrt(d) : s_lift[k]; }, [] {});
    STAMP_P(2);
    // T (rows parked in the lower triangle of B) -> upper triangle of A0; V = T^-1 (parked in the upper triangle of A0) -> B
    for (int e = tid; e < n1 * n1; e += NTD)
    {
      const int i = e % n1, j = e / n1;
      if (i < j)
      {
        const double v = A0[e], t = B[i * n1 + j];
        A0[e] = t;
        B[e] = v;
        A0[i * n1 + j] = 0.0;
        B[i * n1 + j] = 0.0;
      }
      else if (i == j)
      {
        A0[e] = s_part[i];
        B[e] = s_g[i];
      }
    }
    __syncthreads();
  }
  else
  {
  for (int i = tid; i < n1 * n1; i += NTD)
  {
    const int r = i % n1, c = i / n1;
    A0[i] = r <= c ? row_scale * R1[i] : 0.0;  // a factor of 1 row in row_scale^2: the factor of all rows is row_scale x larger
  }
  __syncthreads();
  if (tid < n1)
  {
    double s = 0.0;
    for (int r = 0; r <= tid; ++r) s = fma(A0[tid * n1 + r], A0[tid * n1 + r], s);
    s_norm[tid] = sqrt(s);
  }
  __syncthreads();
  if (tid < n1)
  {
    double mx = 0.0;
    for (int c = 0; c <= tid; ++c) mx = fmax(mx, s_norm[c]);
    const double piv = fabs(A0[tid * n1 + tid]);
    const double floor_k = mx > 0.0 ? 1e-13 * mx : 1.0;  // (nothing but zero columns so far: any diagonal serves)
    const bool residue = !(piv >= floor_k);
    const bool tiny = tid + 1 < n1 && piv < 1e-5 * s_norm[tid];
    s_z[tid] = (residue || tiny) ? 1 : 0;
    s_lift[tid] = residue ? floor_k : piv;
  }
  __syncthreads();
  if (tid == 0)
  {
    int nc = 0;
    for (int c = 0; c < n1; ++c)
      if (!s_z[c]) s_cmap[nc++] = c;
    s_nc = nc;
    for (int c = 0; c < n1; ++c)  // the deferred columns behind the kept ones
      if (s_z[c]) s_cmap[nc++] = c;
    for (int c = 0; c < n1; ++c) s_rend[c] = s_cmap[c] + 1;  // column c of the compacted matrix is zero below its original index
  }
  __syncthreads();
  const int nc = s_nc;
  for (int i = tid; i < n1 * n1; i += NTD) B[i] = A0[s_cmap[i / n1] * n1 + (i % n1)];
  __syncthreads();
  small_qr_lds(B, n1, n1, tid, nc, s_rend);
  // T: kept columns at their positions, deferred columns: coefficients + diagonal
  for (int i = tid; i < n1 * n1; i += NTD) A0[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < nc * nc; i += NTD)
  {
    const int a = i % nc, bcol = i / nc;
    if (a <= bcol) A0[s_cmap[bcol] * n1 + s_cmap[a]] = B[bcol * n1 + a];
  }
  for (int i = tid; i < (n1 - nc) * nc; i += NTD)
  {
    // deferred column k = s_cmap[nc + z]: its coefficient on kept direction a, if that direction's column lies to its left
    const int z = i / (nc > 0 ? nc : 1), a = i - z * nc, k = s_cmap[nc + z];
    if (s_cmap[a] < k) A0[k * n1 + s_cmap[a]] = B[(nc + z) * n1 + a];
  }
  if (tid < n1 && s_z[tid]) A0[tid * n1 + tid] = s_lift[tid];
  __syncthreads();
  }
  STAMP_P(3);
  for (int i = tid; i < n1 * n1; i += NTD) Tout[i] = A0[i];
  if (tid < n1) zmask[tid] = s_z[tid];
  if (!Gs)
  {
    // V = T^-1 (T is nonsingular: full-rank kept block, lifts on the deferred diagonal)
    for (int i = tid; i < n1 * n1; i += NTD) B[i] = 0.0;
    if (tid < n1) s_g[tid] = 1.0 / A0[tid * n1 + tid];
    __syncthreads();
    tri_inverse_lds(A0, s_g, B, n1, tid, nullptr);
  }
  __syncthreads();
  STAMP_P(4);
  for (int i = tid; i < n1 * n1; i += NTD) Vout[i] = B[i];  // T^-1 in natural order: the factor kernel re-evaluates gamma on the norms of ALL rows
  if (tid < n1)
  {
    double g = 0.0;
    for (int i = 0; i <= tid; ++i) g = fma(s_norm[i], fabs(B[tid * n1 + i]), g);
    s_g[tid] = g;
  }
  __syncthreads();
  if (tid < n1)
  {
    double g = 0.0;
    for (int l = 0; l <= tid; ++l) g = fma(s_g[l], fabs(A0[tid * n1 + l]), g);
    s_part[tid] = s_norm[tid] > 0.0 ? g / s_norm[tid] : 0.0;
  }
  __syncthreads();
  if (tid == 0)
  {
    double gamma = 0.0;
    for (int j = 0; j < n1; ++j) gamma = fmax(gamma, s_part[j]);
    // round 1 cannot be accepted above the limit: called off.  Round 0 only has to be worth running as a PRECONDITIONER for round 1 (the
    // factor kernel rejects it as a result by the same figure on all rows): called off when even that is hopeless
    const bool safe = gamma <= (round == 0 ? 1e10 : kCholqrGammaMax);  // (false for NaN)
    if (gamma_out) *gamma_out = gamma;
    if (round == 0) flags[2] = safe ? 1 : 0;
    if (!safe)
    {
      flags[0] = 0;  // the kernels of this round (round 1: flags[0]; round 0: flags[2]) and of the next leave at once
      flags[1] = 1;  // the stand-by runs
    }
  }
  STAMP_P(5);
  // operand order of k_regressor_pgram, in its column space (natural order shifted right by col_shift)
  // (every operand block the consumer loads is written: what lies beyond n1 + col_shift columns is zero, not stale workspace)
  const int nb = nb_w, nt = nb * (nb + 1) / 2;
  for (int i = tid; i < nt * 256; i += NTD)
  {
    const int blk = i >> 8, kk = (i >> 6) & 3, ln = i & 63;
    int cb2 = 0;
    while ((cb2 + 1) * (cb2 + 2) / 2 <= blk) ++cb2;
    const int cb1 = blk - cb2 * (cb2 + 1) / 2;
    const int r = 16 * cb1 + 4 * kk + (ln >> 4) - col_shift, c = 16 * cb2 + (ln & 15) - col_shift;
    W[i] = (r >= 0 && c >= 0 && r < n1 && c < n1 && r <= c) ? B[c * n1 + r] : 0.0;
  }
  STAMP_P(6);
}

// G2 = [G c; c' bb] (the Gram of Q = [A b] W over ALL rows) -> R = chol(G2) T.
//   Columns the preconditioner deferred (zmask): whether they hold anything is decided HERE, on all rows.  Rounding residue of a dependent
//   column reads O(1e-3) in units of its lift (1e-13 x the largest column to its left) and has nothing left after its own elimination
//   either way: its pivot is SKIPPED -- row k of R is exactly zero, the rank deficiency is reported as such.  Anything else is a pivot
//   like any other (a direction the subsample missed, or a nearly dependent one).
//   Accuracy: R = chol(Q'Q) T is as good as cond(Q D^-1)^2 u for the column-equilibrated Q (Cholesky does not see column scales: a
//   preconditioner that is off by a factor per column costs nothing).  The kernel measures it: rho = |Re^-1|_F / sqrt(k) for the factor
//   Re of the equilibrated Gram of the k pivoted columns (1 for orthogonal columns; |Re^-1|_2 <= rho sqrt(k)); rho > 4, or a kept
//   column that turns out dependent on its left neighbours, and the round is not accepted: flags[round] = 1 (flags[0] starts round 1,
//   flags[1] starts the stand-by Householder factorisation).  Round 0 also clears flags[1].
template <bool WIDE>
__global__ __launch_bounds__(NTD) void k_cholqr_factor(const double* __restrict__ G, const double* __restrict__ cvec, const double* __restrict__ bb, int n1,
                                                       int has_b, const double* __restrict__ T_in, const double* __restrict__ V, const int* __restrict__ zmask,
                                                       double* __restrict__ Rout, int* __restrict__ flags, int round, const int* __restrict__ run_flag,
                                                       double* __restrict__ rho_out, double* __restrict__ wide_sq)
{
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* const M = WIDE ? wide_sq : sh;  // [n1][n1] column-major, upper triangle = the running Cholesky factor
  STAMP(0);
  double* const T = M + n1 * n1;          // the identity that becomes the inverse of the Cholesky factor, then T
  __shared__ double s_g0[kMaxN1], s_sc[kMaxN1], s_xd[kMaxN1], s_part[kMaxN1], s_gam[kMaxN1], s_wave[NTD / 64];
  __shared__ int s_flag, s_z[kMaxN1], s_skip[kMaxN1];
  const int tid = threadIdx.x, P = n1 - 1;
  for (int i = tid; i < n1 * n1; i += NTD)
  {
    const int r = i % n1, c = i / n1;
    double v;
    if (r < P && c < P) v = G[(int64_t)c * P + r];
    else if (r == P && c == P) v = bb[0];
    else v = cvec[r < P ? r : c];
    M[i] = r <= c ? v : 0.0;
    T[i] = r == c ? 1.0 : 0.0;
  }
  if (tid < n1) s_z[tid] = zmask[tid] || (!has_b && tid == P);  // no measured torque: the last column is null by construction
  if (tid == 0) s_flag = 0;
  __syncthreads();
  if (tid < n1) s_g0[tid] = M[tid * n1 + tid];  // |Q(:, k)|^2 before anything is eliminated
  // T (global, written by the preconditioner kernel) is needed after the factorisation: the waves that only keep the barrier count
  // during it fetch their share then and hold it in registers (loaded afterwards, the dependent misses cost 8 us)
  static_assert(NTD > 384, "the waves behind the four computing and two parking ones prefetch T");
  constexpr int kWidth = WIDE ? kMaxN1 : kMaxN1Lds;
  constexpr int kTPre = (kWidth * kWidth + (NTD - 384) - 1) / (NTD - 384);
  double tpre[kTPre];
  __syncthreads();
  STAMP(1);
  auto chol = [&](auto... args) {
    if constexpr (WIDE) RDYN_CHOL_WITH_INVERSE<7>(args...);
    else RDYN_CHOL_WITH_INVERSE<6>(args...);
  };
  chol(M, T, n1, tid, s_sc, s_xd, [&](int k, double d) {
    // d / g0 = the squared sine of the angle between Q(:, k) and the columns to its left; "resolved": >= 1e-12
    const bool resolved = d >= 1e-12 * s_g0[k] && s_g0[k] > 0.0;
    const bool large = d >= 0.01;  // the pivot sqrt(d) against the 1/10 mark (in units of the lift for a deferred column)
    int skip = 0, flag = 0;
    if (s_z[k])
    {
      // residue in all rows (in units of the lift)?  Or nothing left after its own elimination: null as far as this round can tell --
      // but if the column was large (its elimination's own rounding, u |Q(:, k)|^2, is above the 1/10 mark), that proves nothing
      skip = !(large && resolved);
      flag = large && !resolved;
    }
    else if (!resolved)
      skip = flag = 1;  // a kept column that turns out to be numerically dependent on its left neighbours in the whole batch
    if (tid == 0)
    {
      s_skip[k] = skip;
      if (flag) s_flag = 1;
    }
    // a skipped pivot: null direction, its Schur complement is rounding residue -- row k of the factor is zero, nothing is eliminated
    return !skip;
  }, [&](int, double d, bool elim) { return elim ? sqrt(d > 1e-30 ? d : 1e-30) : 0.0; }, [&] {
#pragma unroll
    for (int m = 0; m < kTPre; ++m)
    {
      const int e = tid - 384 + (NTD - 384) * m;
      tpre[m] = T_in[e < n1 * n1 ? e : n1 * n1 - 1];  // (unconditional: the loads go out back to back)
    }
  });
  STAMP(2);
  // rho: Re = R2 diag(1 / |Q(:, j)|) over the pivoted columns; Re^-1(i, c) = |Q(:, i)| R2^-1(i, c), R2^-1 parked in the upper triangle of
  // T (zero in the rows / columns of skipped pivots), its diagonal in s_xd
  {
    double part = 0.0;
    for (int e = tid; e < n1 * n1; e += NTD)
    {
      const int i = e % n1, c = e / n1;
      const double x = i < c ? T[e] : (i == c ? s_xd[c] : 0.0);
      part = fma(s_g0[i] * x, x, part);
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) s_wave[tid >> 6] = part;
  }
  __syncthreads();
  // the factor back into the upper triangle: M(i, j) <- row i parked in column i (the lower triangle keeps the parked rows: masked
  // below); T into its place
  {
    const int i = tid & 127;
    if (i < n1)
      for (int j = tid >> 7; j < n1; j += NTD / 128)
      {
        if (i < j) M[j * n1 + i] = M[i * n1 + j];
        else if (i == j) M[j * n1 + i] = s_sc[i];
      }
  }
  if (tid >= 384)
  {
#pragma unroll
    for (int m = 0; m < kTPre; ++m)
    {
      const int e = tid - 384 + (NTD - 384) * m;
      if (e < n1 * n1) T[e] = tpre[m];
    }
  }
  __syncthreads();
  // R = R2 T (upper x upper) on the matrix cores: 16 x 16 tiles of R over the waves, tile (ti, tj) = sum over the column blocks
  // ti .. tj of R2(ti, kb) T(kb, tj), operands straight from LDS (A: lane (cl, g) holds R2(16 ti + cl, k), B: T(k, 16 tj + cl),
  // k = 16 kb + 4 kk + g); tiles below the diagonal are written as zeros
  {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cl = lane & 15, g = lane >> 4;
    const int nbk = (n1 + 15) >> 4;
    for (int tile = wave; tile < nbk * nbk; tile += NTD / 64)
    {
      const int ti = tile % nbk, tj = tile / nbk;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      const int ia = 16 * ti + cl, jb = 16 * tj + cl;
      for (int kb = ti; kb <= tj; ++kb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
        {
          const int k = 16 * kb + 4 * kk + g;
          const bool ka = k < n1 && ia <= k, kt = k < n1 && jb < n1;
          const double a = ka ? M[k * n1 + ia] : 0.0;   // R2(ia, k), zero below the diagonal
          const double b = kt ? T[jb * n1 + k] : 0.0;   // T(k, jb) (its own lower triangle is zero)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
        const int i = 16 * ti + g + 4 * t;  // (the f64 MFMA's own result map: row = (lane >> 4) + 4 reg)
        if (i < n1 && jb < n1) Rout[jb * n1 + i] = acc[t];
      }
    }
  }
  __syncthreads();
  STAMP(3);
  // the growth factor of the round on the column norms of ALL rows (|a_j| = |R(:, j)|; the preconditioner's own figure used the norms
  // of its input factor: the subsample's in round 0, which may say little about the batch).  Three column-wise sums over the rows
  // <= j (R and V from global memory), sixteen lanes per column with their loads in flight together: one thread per column and a
  // dependent load per row took 9-19 us.
  auto column_sums = [&](double* out, auto term) {
    for (int j = tid >> 4; j < n1; j += NTD / 16)
    {
      double sum = 0.0;
#pragma unroll
      for (int m = 0; m < kWidth / 16; ++m)
      {
        const int i = (tid & 15) + 16 * m;
        sum += term(i <= j ? i : j, j, i <= j);
      }
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      sum += __shfl_xor(sum, 8);
      if ((tid & 15) == 0) out[j] = sum;
    }
  };
  column_sums(s_part, [&](int i, int j, bool on) { const double r = Rout[j * n1 + i]; return on ? r * r : 0.0; });
  __syncthreads();
  if (tid < n1) s_part[tid] = sqrt(s_part[tid]);
  __syncthreads();
  column_sums(s_gam, [&](int i, int j, bool on) { return on ? s_part[i] * fabs(V[j * n1 + i]) : 0.0; });  // g_j
  __syncthreads();
  column_sums(s_sc, [&](int l, int j, bool on) { return on ? s_gam[l] * fabs(T[j * n1 + l]) : 0.0; });
  __syncthreads();
  if (tid < n1) s_gam[tid] = s_part[tid] > 0.0 ? s_sc[tid] / s_part[tid] : 0.0;
  __syncthreads();
  STAMP(4);
  if (tid == 0)
  {
    double fro = 0.0;
    int kept = 0;
    for (int w = 0; w < NTD / 64; ++w) fro += s_wave[w];
    for (int c = 0; c < n1; ++c) kept += s_skip[c] ? 0 : 1;
    const double rho = kept > 0 ? sqrt(fro / kept) : 1.0;
    double gamma = 0.0;
    for (int c = 0; c < n1; ++c) gamma = fmax(gamma, s_gam[c]);
    if (!(rho <= 4.0) || !(gamma <= kCholqrGammaMax)) s_flag = 1;
    if (rho_out)
    {
      rho_out[0] = rho;
      rho_out[2] = gamma;
    }
    flags[round] = s_flag;
    if (round == 0) flags[1] = 0;
  }
}

// Factor of the reduced chain -> factor of the chain (rdyn_chain.hpp: [A C b] = [A_red C b] E_aug, E_aug = diag(E, I_K, 1)):
// R = qr(R_red E_aug) by Householder reflections in LDS, one workgroup.  The product has nr = 10 n_red + K + 1 rows and
// n1 = 10 n_joints + K + 1 columns; rows beyond the rank stay zero.  The K component columns (friction_polynomial1.h:126,
// ideal_spring.h:64) belong to INPUT joints, which the reduced chain keeps: they pass through unchanged, like the measured torque.
__global__ __launch_bounds__(NTD) void k_cholqr_expand(const RdynGramExpandArgs a, const double* __restrict__ R_red, double* __restrict__ Rout)
{
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int K = a.n_comp_cols, P = 10 * a.n_joints, n1 = P + K + 1, Pr = 10 * a.n_red, nr = Pr + K + 1;
  const int m = nr;
  double* const B = sh;            // [n1][m] column-major (leading dimension m)
  const int tid = threadIdx.x;
  for (int e = tid; e < n1 * m; e += NTD)
  {
    const int r = e % m, col = e / m;
    double s = 0.0;
    if (col >= P)
    {
      const int cr = Pr + (col - P);  // component column / measured torque: the same column of the reduced factor
      s = r <= cr ? R_red[(int64_t)cr * nr + r] : 0.0;
    }
    else
    {
      const int f = col / 10, p = col - 10 * f, rb = a.red_of[f];
      if (rb >= 0)
        for (int x = 0; x < 10; ++x)
        {
          const int cr = 10 * rb + x;
          if (r <= cr) s = fma(R_red[(int64_t)cr * nr + r], a.X[f * 100 + x * 10 + p], s);
        }
    }
    B[e] = s;
  }
  __syncthreads();
  small_qr_lds(B, m, n1, tid);
  for (int e = tid; e < n1 * n1; e += NTD)
  {
    const int r = e % n1, col = e / n1;
    Rout[e] = (r <= col && r < m) ? B[col * m + r] : 0.0;
  }
}

// R <- qr([R ; R_new]): both n1 x n1 upper triangular, column-major (the accumulate step of the preconditioned route and of the
// expansion above); R_new is zero below its first rows_new rows (the expanded factor of a reduced chain has the reduced chain's row
// count).  Step k: the reflector is [R(k, k); R_new(0 .. k, k)] -- what lies below row k of R is zero and stays zero, and a column
// j > k is touched in R(k, j) and R_new(0 .. k, j) only: no fill-in outside the two triangles.  One barrier per step (every wave forms
// the column norm by itself, the same sum in the same order), four threads per column.
// R_new is PACKED in LDS (column j holds its min(j + 1, rows_new) entries).  R: packed in LDS too (A_GLOBAL = false, n1 <= 136), or
// left where it is and updated in place (A_GLOBAL = true, long chains: every entry of R is read once and written once, by the same
// thread, so nothing travels between threads through it).
template <bool A_GLOBAL>
__global__ __launch_bounds__(NTD) void k_cholqr_fold(const double* __restrict__ R_new, double* R, int n1, int rows_new)
{
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int rn = rows_new < n1 ? rows_new : n1;
  const int tri = n1 * (n1 + 1) / 2;
  auto boff = [&](int j) { return j < rn ? j * (j + 1) / 2 : rn * (rn + 1) / 2 + (j - rn) * rn; };
  auto bcnt = [&](int j) { return j < rn ? j + 1 : rn; };
  double* const A = A_GLOBAL ? R : sh;                 // R: packed A[j (j + 1) / 2 + i] in LDS, or the caller's column-major matrix
  double* const B = A_GLOBAL ? sh : sh + tri;          // R_new, packed
  auto aidx = [&](int i, int j) { return A_GLOBAL ? j * n1 + i : j * (j + 1) / 2 + i; };
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < n1 * n1; e += NTD)
  {
    const int i = e % n1, j = e / n1;
    if (!A_GLOBAL && i <= j) A[j * (j + 1) / 2 + i] = R[e];
    if (i < bcnt(j)) B[boff(j) + i] = R_new[e];
  }
  __syncthreads();
  for (int k = 0; k < n1; ++k)
  {
    const double* const bk = B + boff(k);
    const int cnt = bcnt(k);
    double sigma = 0.0;
    for (int i = lane; i < cnt; i += 64) sigma = fma(bk[i], bk[i], sigma);
    for (int o = 32; o > 0; o >>= 1) sigma += __shfl_xor(sigma, o);
    const double alpha = A[aidx(k, k)];
    double beta = alpha;
    if (sigma > 1e-280)
    {
      const double norm = sqrt(fma(alpha, alpha, sigma));
      beta = alpha > 0.0 ? -norm : norm;
      const double v0 = alpha - beta, scale = 2.0 / fma(v0, v0, sigma);
      const int ncol = n1 - k - 1;
      for (int e = tid; e < (ncol * 4 + NTD - 1) / NTD * NTD; e += NTD)
      {
        const bool on = e < ncol * 4;
        const int j = on ? k + 1 + (e >> 2) : k, q = e & 3;
        double* const akj = A + aidx(k, j);
        double* const bj = B + boff(j);
        const double a_old = (on && q == 0) ? *akj : 0.0;
        double d = v0 * a_old;
        if (on)
          for (int i = q; i < cnt; i += 4) d = fma(bk[i], bj[i], d);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        const double f = scale * d;
        if (on)
        {
          if (q == 0) *akj = fma(-f, v0, a_old);
          for (int i = q; i < cnt; i += 4) bj[i] = fma(-f, bk[i], bj[i]);
        }
      }
    }
    __syncthreads();
    if (tid == 0) A[aidx(k, k)] = beta;
  }
  __syncthreads();
  if (!A_GLOBAL)
    for (int e = tid; e < n1 * n1; e += NTD)
    {
      const int i = e % n1, j = e / n1;
      R[e] = i <= j ? A[j * (j + 1) / 2 + i] : 0.0;
    }
  else
    // the caller's matrix was updated in place: whatever it held below the diagonal is not part of the factor (the LDS variant
    // writes zeros there too)
    for (int e = tid; e < n1 * n1; e += NTD)
      if (e % n1 > e / n1) R[e] = 0.0;
}

template <class K>
hipError_t opt_in_lds_once(K kernel, std::atomic<uint64_t>& done, int max_bytes = 160 * 1024)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  return hipSuccess;
}

template <int NJ, bool ALLREV, int NPAIR, bool WGLOBAL, int XB = 0, int KIN = 0>
hipError_t launch_pgram3(const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds_once(k_regressor_pgram<NJ, ALLREV, NPAIR, WGLOBAL, XB, KIN>, attr);
  if (e != hipSuccess) return e;
  constexpr int NB = (10 * NJ + 1 + 15) / 16 + XB, NT = NB * (NB + 1) / 2;
  size_t lds = (WGLOBAL ? 0 : (size_t)NT * 2048) + (size_t)NPAIR * a.tile_bytes + (KIN ? RDYN_KIN_XCH_BYTES_XV(NJ <= 6 ? 21 : 12) : 0);
  if (lds < (size_t)NT * 2048) lds = (size_t)NT * 2048;  // the final reduction area
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_regressor_pgram<NJ, ALLREV, NPAIR, WGLOBAL, XB, KIN>), dim3(blocks), dim3(128 * NPAIR), lds, st, a, W, run_flag);
  return hipGetLastError();
}
// pairs: 4 = W in LDS beside four tiles; 3 = W in LDS beside three tiles; -4 = four tiles, W read from global memory;
// 2 = two pairs on four SIMDs; 1 = no pairs: every wave sweeps and consumes its own (compact) tile (k_regressor_pgram_solo)
template <int NJ>
hipError_t launch_pgram(const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, int pairs, hipStream_t st)
{
  if (a.n_comp_cols > 0)
  {
    // [Y | C | tau_meas]: chains of up to 6 joints: W beside the four tiles or in global memory; 7 joints: two pairs on four SIMDs
    if constexpr (NJ == 7)
    {
      if (pairs == 2)
        return a.all_revolute ? launch_pgram3<NJ, true, 2, false, 1>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 2, false, 1>(a, W, run_flag, blocks, st);
      if (pairs == -1 || pairs == 1) return rdyn_launch_regressor_pgram_solo(a, W, run_flag, blocks, pairs, st);  // rdyn_pgram_solo.hip
    }
    if constexpr (NJ <= 6)
    {
      if (pairs == 4)
        return a.all_revolute ? launch_pgram3<NJ, true, 4, false, 1>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, false, 1>(a, W, run_flag, blocks, st);
      if (pairs == -4)
        return a.all_revolute ? launch_pgram3<NJ, true, 4, true, 1>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, true, 1>(a, W, run_flag, blocks, st);
    }
    return hipErrorInvalidValue;
  }
  if (a.sweep_lanes)
  {
    // the one-lane-per-sample sweepers: the host built the tile with their padding (rdyn_cholqr_kin_pad) and filled sw_rows for three row waves
    if constexpr (NJ <= 6)
    {
      if (pairs == 4)
        return a.all_revolute ? launch_pgram3<NJ, true, 4, false, 0, 4>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, false, 0, 4>(a, W, run_flag, blocks, st);
    }
    else if (pairs == -4)
      return a.all_revolute ? launch_pgram3<NJ, true, 4, true, 0, 2>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, true, 0, 2>(a, W, run_flag, blocks, st);
    return hipErrorInvalidValue;
  }
  if (pairs == 4)
    return a.all_revolute ? launch_pgram3<NJ, true, 4, false>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, false>(a, W, run_flag, blocks, st);
  if constexpr (NJ >= 7)
  {
    if (pairs == 3)
      return a.all_revolute ? launch_pgram3<NJ, true, 3, false>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 3, false>(a, W, run_flag, blocks, st);
    if (pairs == -4)
      return a.all_revolute ? launch_pgram3<NJ, true, 4, true>(a, W, run_flag, blocks, st) : launch_pgram3<NJ, false, 4, true>(a, W, run_flag, blocks, st);
  }
  return hipErrorInvalidValue;
}
template <int NB>
hipError_t launch_pgram_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, const double* W, double* slabs, const int* run_flag,
                             int blocks, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds_once(k_pgram_rows<NB>, attr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_pgram_rows<NB>), dim3(blocks), dim3(256), (size_t)(NB * (NB + 1) / 2) * 2048, st, A, b, rows, lda, n_cols, W, slabs, run_flag);
  return hipGetLastError();
}
}  // namespace

size_t rdyn_cholqr_w_doubles(int n_joints, int xb)
{
  const int nb = (10 * n_joints + 1 + 15) / 16 + xb;
  return (size_t)(nb * (nb + 1) / 2) * 256;
}

int rdyn_cholqr_pairs(int n_joints, int tile_bytes, int xb)
{
  if (n_joints < 2 || n_joints > 7) return 0;
  const size_t wb = rdyn_cholqr_w_doubles(n_joints, xb) * 8;
  if (n_joints == 7 && xb)
  {
    // -1: k_regressor_pgram_solo, four waves that sweep and consume their own COMPACT tile, W from global memory (the caller checks that
    // four compact tiles fit and asks again with tile_bytes < 0 if they do not); 2: two sweeper / consumer pairs on four SIMDs, W in LDS
    // beside two tiles.  Measured at 14 component columns, N = 1e6 (tools/r4_ab_ident.sh): 3.02 ms (-1), 3.48 (three waves + W in LDS), 3.62 (2)
#ifndef RDYN_CHOLQR_TWO_PAIRS
    if (tile_bytes > 0) return -1;
#endif
    const size_t tb = (size_t)(tile_bytes < 0 ? -tile_bytes : tile_bytes);
    return wb + 2 * tb <= 160 * 1024 ? 2 : 0;
  }
  if (wb + 4 * (size_t)tile_bytes <= 160 * 1024) return 4;
#ifdef RDYN_CHOLQR_W_LDS3
  if (wb + 3 * (size_t)tile_bytes <= 160 * 1024) return 3;
#else
  if (4 * (size_t)tile_bytes <= 160 * 1024) return -4;  // W from global memory (7 joints)
#endif
  return 0;
}

// the one-lane-per-sample sweepers serve pass B of this shape (no component columns yet): the tile padding they use (4; 2 = the compact
// layout at 7 joints), 0 = no.  pairs as returned by rdyn_cholqr_pairs for the PADDED (4) tile.
int rdyn_cholqr_kin_pad(int n_joints, int xb, int pairs)
{
  if (xb) return 0;
  if (n_joints >= 2 && n_joints <= 6) return pairs == 4 ? 4 : 0;
  return (n_joints == 7 && pairs == -4) ? 2 : 0;
}

// nb = ceil(n1 / 16) <= 6 column blocks (n1 = n_cols + (b != null)); slabs: [blocks][nb (nb + 1) / 2 * 256]
hipError_t rdyn_launch_pgram_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, const double* W, double* slabs,
                                  const int* run_flag, int blocks, hipStream_t st)
{
  switch ((n_cols + (b ? 1 : 0) + 15) / 16)
  {
  case 1: return launch_pgram_rows<1>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 2: return launch_pgram_rows<2>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 3: return launch_pgram_rows<3>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 4: return launch_pgram_rows<4>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 5: return launch_pgram_rows<5>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 6: return launch_pgram_rows<6>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  case 7: return launch_pgram_rows<7>(A, b, rows, lda, n_cols, W, slabs, run_flag, blocks, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t rdyn_launch_regressor_pgram(int n_joints, const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, int pairs, hipStream_t st)
{
  switch (n_joints)
  {
  case 2: return launch_pgram<2>(a, W, run_flag, blocks, pairs, st);
  case 3: return launch_pgram<3>(a, W, run_flag, blocks, pairs, st);
  case 4: return launch_pgram<4>(a, W, run_flag, blocks, pairs, st);
  case 5: return launch_pgram<5>(a, W, run_flag, blocks, pairs, st);
  case 6: return launch_pgram<6>(a, W, run_flag, blocks, pairs, st);
  case 7: return launch_pgram<7>(a, W, run_flag, blocks, pairs, st);
  default: return hipErrorInvalidValue;
  }
}

int rdyn_cholqr_max_cols() { return kMaxN1; }          // rdyn_tsqr on a materialised matrix (the squares of the dense steps in the workspace beyond 96)
int rdyn_cholqr_max_cols_lds() { return kMaxN1Lds; }  // the fused regressor routes (pass B holds six column blocks)

hipError_t rdyn_launch_cholqr_precond(const double* R1, const double* Gs, const double* cs, const double* bbs, int n1, int col_shift, int nb_w,
                                      double row_scale, double* T, double* W, double* V, int* zmask, int* flags, int round, const int* run_flag,
                                      double* gamma_out, hipStream_t st, double* wide_sq)
{
  if (n1 < 1 || n1 > kMaxN1 || 16 * nb_w < n1 + col_shift) return hipErrorInvalidValue;
  if (n1 > kMaxN1Lds)
  {
    if (!wide_sq) return hipErrorInvalidValue;  // (2 n1^2 doubles of the caller's workspace)
    hipLaunchKernelGGL(k_cholqr_precond<true>, dim3(1), dim3(NTD), 0, st, R1, Gs, cs, bbs, n1, col_shift, nb_w, row_scale, T, W, V, zmask, flags, round, run_flag, gamma_out, wide_sq);
    return hipGetLastError();
  }
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds_once(k_cholqr_precond<false>, attr, 146 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cholqr_precond<false>, dim3(1), dim3(NTD), ((size_t)2 * n1 * n1 + n1) * sizeof(double), st, R1, Gs, cs, bbs, n1, col_shift, nb_w, row_scale, T, W, V, zmask, flags, round, run_flag,
                     gamma_out, nullptr);
  return hipGetLastError();
}

hipError_t rdyn_launch_cholqr_factor(const double* G, const double* c, const double* bb, int n1, int has_b, const double* T, const double* V, const int* zmask,
                                     double* R, int* flags, int round, const int* run_flag, double* rho_out, hipStream_t st, double* wide_sq)
{
  if (n1 < 2 || n1 > kMaxN1 || round < 0 || round > 1) return hipErrorInvalidValue;
  if (n1 > kMaxN1Lds)
  {
    if (!wide_sq) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_cholqr_factor<true>, dim3(1), dim3(NTD), 0, st, G, c, bb, n1, has_b, T, V, zmask, R, flags, round, run_flag, rho_out, wide_sq);
    return hipGetLastError();
  }
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds_once(k_cholqr_factor<false>, attr, 146 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cholqr_factor<false>, dim3(1), dim3(NTD), (size_t)2 * n1 * n1 * sizeof(double), st, G, c, bb, n1, has_b, T, V, zmask, R, flags, round, run_flag, rho_out, nullptr);
  return hipGetLastError();
}

// a.G_red / c_red / bb_red / G / c / bb are unused here: the factors travel as separate arguments.  R must not alias R_red.
hipError_t rdyn_launch_cholqr_expand(const RdynGramExpandArgs& a, const double* R_red, double* R, hipStream_t st)
{
  const int n1 = 10 * a.n_joints + a.n_comp_cols + 1, nr = 10 * a.n_red + a.n_comp_cols + 1;
  const size_t lds = ((size_t)n1 * nr + nr) * sizeof(double);
  if (lds > 156 * 1024) return hipErrorInvalidValue;
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds_once(k_cholqr_expand, attr, 156 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cholqr_expand, dim3(1), dim3(NTD), lds, st, a, R_red, R);
  return hipGetLastError();
}

size_t rdyn_cholqr_expand_lds_bytes(int n_joints, int n_red, int n_comp_cols)
{
  return ((size_t)(10 * n_joints + n_comp_cols + 1) * (10 * n_red + n_comp_cols + 1) + (10 * n_red + n_comp_cols + 1)) * sizeof(double);
}

hipError_t rdyn_launch_cholqr_fold(const double* R_new, double* R, int n1, hipStream_t st, int rows_new)
{
  if (n1 < 1) return hipErrorInvalidValue;
  const int rn = (rows_new > 0 && rows_new < n1) ? rows_new : n1;
  const size_t b_doubles = (size_t)rn * (rn + 1) / 2 + (size_t)(n1 - rn) * rn;
  const size_t both = ((size_t)n1 * (n1 + 1) / 2 + b_doubles) * sizeof(double);
  if (n1 <= kMaxFoldN1 && both <= 156 * 1024)
  {
    static std::atomic<uint64_t> attr{0};
    hipError_t e = opt_in_lds_once(k_cholqr_fold<false>, attr, 156 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cholqr_fold<false>, dim3(1), dim3(NTD), both, st, R_new, R, n1, rn);
    return hipGetLastError();
  }
  if (b_doubles * sizeof(double) > 156 * 1024) return hipErrorInvalidValue;
  static std::atomic<uint64_t> attr_g{0};
  hipError_t e = opt_in_lds_once(k_cholqr_fold<true>, attr_g, 156 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cholqr_fold<true>, dim3(1), dim3(NTD), b_doubles * sizeof(double), st, R_new, R, n1, rn);
  return hipGetLastError();
}

// columns the consumer of k_regressor_pgram keeps in front of the natural order (the padding of its last 16-column block)
int rdyn_cholqr_col_shift(int n_joints, int xb) { return xb ? 0 : 16 * ((10 * n_joints + 1 + 15) / 16) - (10 * n_joints + 1); }
