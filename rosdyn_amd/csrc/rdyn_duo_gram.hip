// rdyn_duo_gram.hip -- regressor rows -> fp64-MFMA Gram with the two instruction streams on TWO co-resident waves.
//
// rdyn_pipe_gram.hip interleaves the fp64 VALU sweep and the fp64 MFMA k-steps inside ONE wave.  Measured on gfx950
// (tools/fp64_issue.hip, profiles/r2/fp64_issue.txt): a lone wave pays ~5 cycles per fp64 VALU instruction, 83 per
// v_mfma_f64_16x16x4_f64 and ~80 more every time its stream switches between the two, and every scalar / LDS / 32-bit
// instruction of the sweep (40 % of its stream) takes an issue slot of its own: 31 k cycles per 16-sample tile.
// Here a 512-thread workgroup holds four PAIRS of waves (wave p and wave p + 4 land on the same SIMD: a workgroup's waves are
// dealt to the SIMDs cyclically):
//   * the SWEEPER (waves 0-3) runs the row-pair forward sweep of a 16-sample tile (4 lanes per sample, 2 regressor rows per
//     lane, ~140 registers, no accumulators) and drops each finished link's 10-vectors into the pair's LDS tile;
//   * the CONSUMER (waves 4-7) holds the Gram accumulators (80-120 registers) and runs the MFMA k-steps of the PREVIOUS tile
//     out of the same LDS buffer: row group j (the 16 samples of input joint j) is read while the sweeper computes link j of
//     the next tile, and only then -- after a workgroup barrier -- does the sweeper overwrite the columns of link j, which
//     no later group reads (group j' > j only touches columns of links >= j').
// One s_barrier per link plus one per tile keeps the two in step; the hardware interleaves the two streams cycle by cycle,
// each role has its own register budget (the kernel fits 256 registers = two waves per SIMD), and scalar / LDS / address
// instructions of one wave issue under the other's fp64 work.  Same tile layout, operand order and epilogue arithmetic as
// rdyn_lds_gram.hip / rdyn_pipe_gram.hip: results are bit-identical to them.
// Requirements (else rdyn_regressor_gram uses the single-wave kernels): 2 <= chain joints <= 7, input joints in chain order,
// four tiles inside 160 KB of LDS.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_gram_common.h"
#include "rdyn_duo_common.h"

#ifndef RDYN_DUO_SWEEP_UNROLL
#define RDYN_DUO_SWEEP_UNROLL 1  // link loop of the sweeper: 1 = rolled (unrolled measured no better: more moves and SGPR spills)
#endif

namespace
{

// all 8 waves; LDS traffic of this wave retired first.  Plain s_barrier (not __syncthreads): no vmcnt(0), so the sweeper's
// global prefetch of the next tile's inputs stays in flight across the barriers.
#ifdef RDYN_DUO_STAMPS  // diagnostic build only (tools/kbench KB_STAMPS=1): cycles every wave spends inside the barriers
// (+ a timeline: workgroup 0, trip RDYN_DUO_STAMP_TRIP, every wave's clock before and after each of its barriers -> slab slot 600)
#ifndef RDYN_DUO_STAMP_TRIP
#define RDYN_DUO_STAMP_TRIP 7
#endif
#define DUO_STAMP_DECL unsigned long long st_wait = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_tmp; long long st_it = -1; int st_n = 0
#define DUO_STAMP_TRIP(it_) st_it = (it_)
#define DUO_STAMP_MARK(t_) do { if (blockIdx.x == 0 && st_it == RDYN_DUO_STAMP_TRIP && (threadIdx.x & 63) == 0 && st_n < 32) fa.slabs[(int64_t)600 * 4096 + (threadIdx.x >> 6) * 32 + st_n++] = (double)((t_) - st_t0); } while (0)
#define DUO_STAMP_HERE(wait_lds) do { if (wait_lds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); DUO_STAMP_MARK(__builtin_amdgcn_s_memtime()); } while (0)
#define DUO_BARRIER_LDS() do { st_tmp = __builtin_amdgcn_s_memtime(); DUO_STAMP_MARK(st_tmp); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); const unsigned long long st_e = __builtin_amdgcn_s_memtime(); st_wait += st_e - st_tmp; DUO_STAMP_MARK(st_e); } while (0)
#define DUO_BARRIER() do { st_tmp = __builtin_amdgcn_s_memtime(); DUO_STAMP_MARK(st_tmp); asm volatile("s_barrier" ::: "memory"); const unsigned long long st_e = __builtin_amdgcn_s_memtime(); st_wait += st_e - st_tmp; DUO_STAMP_MARK(st_e); } while (0)
#define DUO_STAMP_OUT(NT_) do { if (lane == 0) { double* o = fa.slabs + ((int64_t)(256 + blockIdx.x) * (NT_ * 256)) + wave * 2; o[0] = (double)(__builtin_amdgcn_s_memtime() - st_t0); o[1] = (double)st_wait; } } while (0)
#else
#define DUO_STAMP_DECL
#define DUO_STAMP_TRIP(it_)
#define DUO_STAMP_HERE(wait_lds)
#define DUO_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DUO_BARRIER() asm volatile("s_barrier" ::: "memory")
#define DUO_STAMP_OUT(NT_)
#endif

// XB: extra 16-column blocks for the component columns of rdyn_identification_gram (0: plain regressor Gram)
typedef double d4h __attribute__((ext_vector_type(4), aligned(16)));  // operand quads: 16-byte aligned in the compact tile layout

// ALLREV: every chain joint is revolute (the UR / Panda arms of BASELINE.json): the sweeper drops the joint-kind selects and the
// prismatic terms (14 % of its instructions; only instantiated with DIRECT)
// KIN != 0: the one-lane-per-sample sweepers (DIRECT only; KIN = the padding of the tile columns in doubles, 4 or 2): sweeper wave 0
// runs the link kinematics of the workgroup's 64 samples once, waves 1-3 the regressor rows -- the consumers do not see the difference
template <int NJ, bool DIRECT, int XB, bool ALLREV, int KIN = 0>
__global__ __launch_bounds__(KIN ? 768 : 512) void k_regressor_gram_duo(const RdynLdsGramArgs fa)
{
  constexpr int NB = (10 * NJ + 1 + 15) / 16 + XB;
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int P = 10 * NJ;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // KIN: TWELVE waves -- eight sweepers (wave 0 the kinematics, waves 1-7 the rows: the sweeper waves' own chain between two barriers
  // is what a tile costs, so it is split over as many waves as the register file holds: three per SIMD, 168 registers each) and the
  // four consumers behind them
  constexpr int NSW = KIN ? 8 : 4;
  const int pair = KIN ? (wave - NSW) & 3 : wave & 3;
  const bool sweeper = wave < NSW;
  char* const tile = lds_raw + (size_t)pair * fa.tile_bytes;  // shared by the pair
  const int n = fa.n_active;
  // tile_stride > 1: only every tile_stride-th 16-sample tile (the subsample pass of the robust factor, rdyn_cholqr.hip)
  const int64_t t_mul = fa.tile_stride > 1 ? fa.tile_stride : 1;
  const int64_t n_tiles = ((fa.n_samples + 15) / 16 + t_mul - 1) / t_mul;
  const int64_t t_step = (int64_t)gridDim.x * 4;
  const int64_t t_first = (int64_t)blockIdx.x * 4 + pair;
  // same trip count for every wave of the workgroup (the barriers are workgroup-wide): pairs without a tile sweep masked samples
  const int64_t trips_raw = (n_tiles - (int64_t)blockIdx.x * 4 + t_step - 1) / t_step;
  const int64_t trips = trips_raw > 0 ? trips_raw : 0;
  DUO_STAMP_DECL;

  if (sweeper && KIN)
  {
    // ================================================================ one lane per sample: 64 samples, the link kinematics ONCE
    // (rdyn_kin_sweepers.inc: wave 0 the kinematics, waves 1-7 the rows, at most two each).  Doubles per sample and exchange buffer:
    // 30 (R, the joint offset, w, al, d, the b-matrix) with component columns, 21 without (the row waves have the slack to rebuild the
    // b-matrix: 9 stores and ~18 instructions less on the kinematics wave, the longest chain of an interval: 515 -> 487 us at 6
    // joints; with component columns the row waves are the longer side), 12 at 7 joints (what the compact tiles leave of the LDS).
    constexpr int XV = NJ <= 6 ? (XB > 0 ? 30 : 21) : 12;
    constexpr int KIN_SLOTS = 2;
    const int KIN_SW = wave;
    char* const kin_tiles = lds_raw;
    double* const kin_xch_base = (double*)(lds_raw + (size_t)4 * fa.tile_bytes);
#include "rdyn_kin_sweepers.inc"
  }
  else if (sweeper)
  {
    // ================================================================ sweeper: 16 samples x 4 lanes, rows 2k, 2k + 1
    ChainPtr c = as_const(fa.chain);
#ifdef RDYN_DUO_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(RDYN_DUO_SWEEP_PRIO);
#endif
    const int s_loc = lane >> 2, k = lane & 3;
    // lane k of a sample's quad owns regressor rows k (slot 0) and k + 4 (slot 1) -- the same split as the inputs it fetches.
    // Rows >= 4 belong to joints that sit at chain index >= fB: before link fB slot 1 is identically zero and is skipped
    // (wave-uniform), which removes a third of the row work of a 6-joint chain.
    const int r0 = k, r1 = k + 4;
    RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
    int fB = DIRECT ? 4 : NJ;
    if (!DIRECT)
      for (int f = NJ - 1; f >= 0; --f)
        if (fa.lds_m[f] >= 5) fB = f;
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
      // a masked sample keeps zero joint twists for both rows (its rows never "start"), so every regressor entry is 0
      const int m0idx = valid ? r0 : -2, m1idx = valid ? r1 : -2;
      const double qa = nqa, dqa = ndqa, ddqa = nddqa, qb = nqb, dqb = ndqb, ddqb = nddqb;
      const double tb0 = valid ? nb0 : 0.0, tb1 = valid ? nb1 : 0.0;
      if (tl + t_step < n_tiles) fetch(tl + t_step);  // in flight during this tile's sweep
      // sin / 1 - cos of MY two input joints, once per tile (the two evaluations interleave); the link loop gets its joint's
      // pair by a quad shuffle instead of every lane of the quad repeating the same sincos per link
      double sna, csa, snb, csb;
      rdyn_sincos(qa, &sna, &csa);
      rdyn_sincos(qb, &snb, &csb);
      const double oca = 1.0 - csa, ocb = 1.0 - csb;

      V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
      V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
      V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);
      if constexpr (DIRECT)
      {
#pragma unroll
        for (int f = 0; f < NJ; ++f)
        {
#include "rdyn_duo_link_body.inc"
        }
      }
      else
      {
#pragma unroll 1
        for (int f = 0; f < NJ; ++f)
        {
#include "rdyn_duo_link_body.inc"
        }
      }
      if (XB > 0 || fa.n_comps > 0)
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
    // ================================================================ consumer: MFMA k-steps of the previous tile
#ifdef RDYN_DUO_MFMA_PRIO
    __builtin_amdgcn_s_setprio(RDYN_DUO_MFMA_PRIO);
#endif
    const int cl = lane & 15, g = lane >> 4;
    // per column block: LDS offset of MY column (for row group 0) and the row groups [collo, colm) it stores
    int colbase[NB], colm[NB], collo[NB];
    const int K = XB > 0 ? fa.n_comp_cols : 0;  // component columns: in FRONT of the accumulation order (a constant 0 without them)
    const int KF = K;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
    {
      // The Gram is accumulated in the column order [component columns | tau_meas | link NJ-1 | ... | link 0]: row group j is non-zero
      // in the FIRST KF + 10 (NJ - j) + 1 columns (its own joint's component columns among the first KF), so its zero band ends at a
      // 16-column boundary more often than in natural order (without components: 48 instead of 62 tile k-steps per tile at 7 joints,
      // 33 instead of 36 at 6; with 14 / 12 of them: 76 instead of 94, 50 instead of 59); k_gram_finish undoes the permutation
      const int pp = 16 * cb + cl;
      const int pl = pp - KF - 1;  // position among the link columns
      const int p = pp < KF ? P + pp : (pp == KF ? P + K : (pl >= P ? P + K + 1 : 10 * (NJ - 1 - pl / 10) + pl % 10));
      int base = 0, hi = 0, lo = 0;
      if (p < P)
      {
        const int f = p / 10;
        base = fa.lds_off[f] + (p - 10 * f) * fa.lds_stride[f];
        hi = fa.lds_m[f];
      }
      else if (p < P + K)
      {
        lo = fa.comp_col_row[p - P];          // one row group: the component's own joint
        hi = lo + 1;
        base = fa.lds_off_c + (p - P) * fa.comp_stride - lo * 128;
      }
      else if (p == P + K)
      {
        base = fa.lds_off_b;
        hi = n;
      }
      colbase[cb] = base + g * 32;
      colm[cb] = hi;
      collo[cb] = lo;
    }
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    // operands of row group j: column blocks <= band = (KF + 10 (NJ - j)) >> 4 (input joints in chain order: joint j sits at chain
    // index >= j; a run-time figure with component columns, folded per unrolled group without)
    auto lds_group = [&](int j, int band, d4* op) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
      {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
        if (cb <= band && j < colm[cb] && j >= collo[cb]) x = *(const d4h*)(tile + colbase[cb] + j * 128);
        op[cb] = x;
      }
    };
    auto mfma_band = [&](const d4* op, int band) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
        int ti = 0;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
          for (int rb = 0; rb <= cb; ++rb)
          {
#ifdef RDYN_DUO_4X4_TIMING  // timing experiment only (wrong numbers): the issue cost of four (three on the diagonal) 4x4x4_4b MFMAs per tile k-step
            if (cb <= band)
            {
              acc[ti][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[rb][t], op[cb][t], acc[ti][0], 0, 0, 0);
              acc[ti][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[rb][t], op[cb][t], acc[ti][1], 0, 0, 0);
              acc[ti][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[rb][t], op[cb][t], acc[ti][2], 0, 0, 0);
              if (rb != cb) acc[ti][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(op[rb][t], op[cb][t], acc[ti][3], 0, 0, 0);
            }
#else
            if (cb <= band) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rb][t], op[cb][t], acc[ti], 0, 0, 0);
#endif
            ++ti;
          }
      }
    };
    // 7 joints + component columns: 21 accumulator tiles (168 registers) + two operand sets (96) do not fit 256 registers (468 B of
    // scratch in the hot loop): ONE operand set there -- the next group's loads are issued behind this group's MFMAs instead of in front
    constexpr bool ONEBUF = (NJ >= 7 && XB > 0) || KIN != 0;  // (KIN: three waves per SIMD, 168 registers)
    d4 opa[NB], opb[ONEBUF ? 1 : NB];
    if (KIN) DUO_BARRIER();  // the prologue of the one-lane-per-sample sweepers (link 0 of the first tile is published)
    for (int64_t it = 0; it <= trips; ++it)
    {
      DUO_STAMP_TRIP(it);
      // it > 0: the tile in LDS is complete (nothing to consume while the first tile is being swept): group 0
      const bool have = it > 0;
      if (have) lds_group(0, (KF + 10 * NJ) >> 4, opa);
#pragma unroll
      for (int f = 0; f < NJ; ++f)
      {
        d4* const cur = (ONEBUF || !(f & 1)) ? opa : opb;
        d4* const nxt = (ONEBUF || (f & 1)) ? opa : opb;
        if (it < trips) DUO_BARRIER_LDS();  // my reads of group f have returned -> the sweeper may overwrite link f's columns
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (have)
        {
          if constexpr (ONEBUF)
          {
            mfma_band(cur, (KF + 10 * (NJ - f)) >> 4);
            if (f + 1 < NJ) lds_group(f + 1, (KF + 10 * (NJ - f - 1)) >> 4, nxt);
          }
          else
          {
            if (f + 1 < NJ) lds_group(f + 1, (KF + 10 * (NJ - f - 1)) >> 4, nxt);
            mfma_band(cur, (KF + 10 * (NJ - f)) >> 4);
          }
        }
      }
      if (it < trips) DUO_BARRIER_LDS();  // end of the sweeper's tile
    }

    DUO_STAMP_OUT(NT);
    // ---- block reduction of the four consumers (fixed order), this block's Gram slab.  The reduction area overlays the
    // tiles: every consumer must be done with its own first.
    DUO_BARRIER_LDS();
    double* red = (double*)lds_raw;
    const int cw = wave - NSW;
    for (int w = 0; w < 4; ++w)
    {
      if (cw == w)
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
  {
    DUO_STAMP_OUT(NT);
    // the sweepers take part in the five barriers of the reduction above
    for (int w = 0; w < 5; ++w) __builtin_amdgcn_s_barrier();
  }
  asm volatile("" ::: "memory");
  {
    const double* red = (const double*)lds_raw;
    double* slab = fa.slabs + (int64_t)blockIdx.x * (NT * 256);
    for (int i = threadIdx.x; i < NT * 256; i += (KIN ? 768 : 512)) slab[i] = red[i];
  }
}

template <int NJ, bool DIRECT, int XB, bool ALLREV = false, int KIN = 0>
hipError_t launch_duo_nj2(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)k_regressor_gram_duo<NJ, DIRECT, XB, ALLREV, KIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_regressor_gram_duo<NJ, DIRECT, XB, ALLREV, KIN>), dim3(blocks), dim3(KIN ? 768 : 512), lds_bytes, st, a);
  return hipGetLastError();
}
template <int NJ>
hipError_t launch_duo_nj(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  // direct = every chain joint is an input joint, in chain order (the tile layout tables then follow from NJ alone)
  bool direct = a.n_active == NJ;
  for (int f = 0; direct && f < NJ; ++f) direct = a.lds_m[f] == f + 1 && a.first_col[f] == 10 * f && a.lds_stride[f] == (16 * (f + 1) + 4) * 8;
  // component columns always take the XB = 1 instantiation (one more column block, the component columns in front of the order)
  const int xb = a.n_comp_cols > 0 ? 1 : 0;
  if (a.sweep_lanes)
  {
    // one lane per sample: the link kinematics once per workgroup and 64 samples, the rows on three waves.  The exchange area sits
    // behind the four tiles (the host sized lds_bytes for it); 7 joints: the compact tile layout (2 doubles of column padding).
    bool compact = a.n_active == NJ;
    for (int f = 0; compact && f < NJ; ++f) compact = a.lds_m[f] == f + 1 && a.first_col[f] == 10 * f && a.lds_stride[f] == (16 * (f + 1) + 2) * 8;
    if constexpr (NJ <= 6)
    {
      if (direct && xb == 0) return a.all_revolute ? launch_duo_nj2<NJ, true, 0, true, 4>(a, blocks, lds_bytes, st) : launch_duo_nj2<NJ, true, 0, false, 4>(a, blocks, lds_bytes, st);
      if constexpr (NJ >= 5)
        if (direct && xb == 1) return a.all_revolute ? launch_duo_nj2<NJ, true, 1, true, 4>(a, blocks, lds_bytes, st) : launch_duo_nj2<NJ, true, 1, false, 4>(a, blocks, lds_bytes, st);
    }
    else if (compact && xb == 0)
      return a.all_revolute ? launch_duo_nj2<NJ, true, 0, true, 2>(a, blocks, lds_bytes, st) : launch_duo_nj2<NJ, true, 0, false, 2>(a, blocks, lds_bytes, st);
    return hipErrorInvalidValue;  // (the host asks for these sweepers only where they exist)
  }
  if (xb == 0 && direct && a.all_revolute) return launch_duo_nj2<NJ, true, 0, true>(a, blocks, lds_bytes, st);
  if (xb == 0) return direct ? launch_duo_nj2<NJ, true, 0>(a, blocks, lds_bytes, st) : launch_duo_nj2<NJ, false, 0>(a, blocks, lds_bytes, st);
  if constexpr (NJ >= 5)  // identification with component columns: one extra column block, arms of 5-7 joints
  {
    if (xb == 1 && direct && a.all_revolute) return launch_duo_nj2<NJ, true, 1, true>(a, blocks, lds_bytes, st);
    if (xb == 1) return direct ? launch_duo_nj2<NJ, true, 1>(a, blocks, lds_bytes, st) : launch_duo_nj2<NJ, false, 1>(a, blocks, lds_bytes, st);
  }
  return hipErrorInvalidValue;
}
}  // namespace

bool rdyn_regressor_gram_duo_supported(int n_cols) { return n_cols >= 20 && n_cols <= 70; }
int rdyn_regressor_gram_duo_kin_pad(int n_joints, int n_comp_cols)
{
  if (n_joints >= 2 && n_joints <= 6) return (n_comp_cols == 0 || n_joints >= 5) ? 4 : 0;
  return (n_joints == 7 && n_comp_cols == 0) ? 2 : 0;
}
bool rdyn_regressor_gram_duo_supports_components(int n_cols, int n_comp_cols)
{
  if (!rdyn_regressor_gram_duo_supported(n_cols) || n_comp_cols < 0 || n_comp_cols > 96) return false;
  if (n_comp_cols == 0) return true;
  const int nb1 = (n_cols + 1 + 15) / 16 + 1;  // one extra 16-column block
  return n_cols >= 50 && n_cols + n_comp_cols + 1 <= 16 * nb1;
}

hipError_t rdyn_launch_regressor_gram_duo(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  switch (n_cols / 10)  // chain joints
  {
  case 2: return launch_duo_nj<2>(a, blocks, lds_bytes, st);
  case 3: return launch_duo_nj<3>(a, blocks, lds_bytes, st);
  case 4: return launch_duo_nj<4>(a, blocks, lds_bytes, st);
  case 5: return launch_duo_nj<5>(a, blocks, lds_bytes, st);
  case 6: return launch_duo_nj<6>(a, blocks, lds_bytes, st);
  case 7: return launch_duo_nj<7>(a, blocks, lds_bytes, st);
  default: return hipErrorInvalidValue;
  }
}
