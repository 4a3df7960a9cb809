// rdyn_pgram_solo.hip -- pass B of the preconditioned R factor (rdyn_cholqr.hip) for a 7-joint arm WITH component columns: the
// kernel whose waves sweep and consume their own tile (its own translation unit: one instantiation per quantised column shift).
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
#define RDYN_CHOLQR_AHEAD 2  // rows of W operands requested ahead of their MFMAs (as in rdyn_cholqr.hip)
#endif
#define DUO_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define DUO_BARRIER() asm volatile("s_barrier" ::: "memory")

namespace
{

typedef const __attribute__((address_space(1))) double* GlobalD;

// pass B for a 7-joint arm WITH component columns (rdyn_identification_tsqr on a Panda-class arm with friction): 21 accumulator tiles
// + 6 product tiles do not fit the 256 registers of a wave that shares its SIMD with its sweeper.  fp64 MFMA and fp64 VALU exclude each
// other on a SIMD cycle for cycle (profiles/r4/duo_gram_ab.txt: a tile costs the SUM of the two streams whoever issues them), so
// nothing is lost when ONE wave does both: four waves per workgroup, one per SIMD (up to 512 registers each), every wave sweeps a
// 16-sample tile into ITS OWN LDS tile and then multiplies and accumulates it -- no pairing, no workgroup barrier in the loop, and all
// four SIMDs issue MFMAs (the two-pair version kept the matrix pipe of two SIMDs idle: 3.62 ms per 1e6 samples against 3.02 here -- a lone
// wave hides none of its own latencies: 74 k cycles per tile against the 52 k of the two streams' sum).
//   tile      the COMPACT layout (2 doubles of padding per column: four tiles of 39 KB fit 160 KB), DIRECT sweeper
//   W         from global memory (43 KB: L1 / L2 resident), rows requested two ahead of their MFMAs
//   columns   natural order shifted right by fa.col_shift = 96 - (71 + K): the padding joins every row group's zero band
//             (floor((shift + 10 j) / 16) whole blocks skipped: 608 instead of 752 MFMAs per tile at K = 14); the band is a run-time
//             figure, each row group dispatches to the code of its band.
#define RDYN_SOLO_PAD 2
// NW waves per workgroup; WGLOBAL: W read from global memory (NW = 4: four compact tiles fill the LDS) or from LDS beside NW = 3 tiles.
// SHC >= 0: fa.col_shift is this compile-time figure -- every row group's zero band is then known to the compiler: no dispatch per row
// group, no copies of the accumulators where the dispatch's branches meet (631 of them per tile), loads hoisted across row groups:
// 2.76 -> 2.11 ms per 1e6 samples at K = 14.  The host quantises the shift to the instantiated values (rdyn_cholqr_solo_col_shift); -1:
// any shift, dispatched per row group at run time.
template <int NJ, bool ALLREV, int NW, bool WGLOBAL, int SHC>
__global__ __launch_bounds__(64 * NW) void k_regressor_pgram_solo(const RdynLdsGramArgs fa, const double* __restrict__ Wg, const int* __restrict__ run_flag)
{
  constexpr bool DIRECT = true;
  constexpr int XB = 1, NB = (10 * NJ + 1 + 15) / 16 + XB, NT = NB * (NB + 1) / 2, P = 10 * NJ, PAD = RDYN_SOLO_PAD;
  constexpr int WB = WGLOBAL ? 0 : NT * 2048;
  if (run_flag && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (!WGLOBAL)
  {
    double* const wlds = (double*)lds_raw;
    for (int i = threadIdx.x; i < NT * 256; i += 64 * NW) wlds[i] = Wg[i];
    __syncthreads();
  }
  char* const tile = lds_raw + WB + (size_t)wave * fa.tile_bytes;
  const int n = fa.n_active;
  const int SH = SHC >= 0 ? SHC : fa.col_shift, K = fa.n_comp_cols;
  const int64_t n_tiles = (fa.n_samples + 15) / 16;
  const int64_t t_step = (int64_t)gridDim.x * NW, t_first = (int64_t)blockIdx.x * NW + wave;
  // ---------------- sweeper state (as in k_regressor_pgram)
  ChainPtr c = as_const(fa.chain);
  const int s_loc = lane >> 2, k = lane & 3;
  const int r0 = k, r1 = k + 4;
    RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
  const int fB = 4;
  double nqa = 0.0, ndqa = 0.0, nddqa = 0.0, nqb = 0.0, ndqb = 0.0, nddqb = 0.0, nb0 = 0.0, nb1 = 0.0;
  auto fetch = [&](int64_t tile_index) {
    int64_t sx = tile_index * 16 + s_loc;
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
  // ---------------- consumer state
  const int cl = lane & 15, g = lane >> 4;
  // operand ids (4 cb1 + kk) from X0 on may reach the columns behind the links ([C (K) | tau_meas | padding]): per lane offset and row
  // group, looked up once (-3: still a link column; -1: every row group (tau_meas); -2: nothing)
  constexpr int X0 = P / 4, NX = 4 * NB - X0;
  int xoff[NX], xrow[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i)
  {
    const int col = 4 * (X0 + i) + g - SH;
    int off = 0, row = -2;
    if (col < P) row = -3;
    else if (col < P + K)
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
  auto link_operand = [&](int col, int j) -> double {
    if (col < 0) return 0.0;  // the padding in front
    const int f = (col * 205) >> 11;  // col / 10 for col < 1024
    const int off = f * (640 * f + 640 + 80 * PAD) + (col - 10 * f) * (128 * f + 128 + 8 * PAD);
    return j <= f ? *(const double*)(tile + off + cl * 8 + j * 128) : 0.0;
  };
  auto a_operand = [&](int cb1, int kk, int j) -> double {
    const int id = 4 * cb1 + kk;
    if (id < X0) return link_operand(4 * id + g - SH, j);
    const int i = id - X0;
    if (xrow[i] == -3) return link_operand(4 * id + g - SH, j);
    return (xrow[i] == -1 || xrow[i] == j) ? *(const double*)(tile + xoff[i] + j * 128) : 0.0;
  };
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  GlobalD wgp = (GlobalD)Wg;  // (a global pointer by type: see k_regressor_pgram)
  auto wglob = [&](int blk_kk) -> double { return (wgp + blk_kk * 64)[lane]; };
  // one row group (the 16 samples of joint j) with the first BAND column blocks in its zero band: Q = X W, then acc += Q'Q
  auto row_group = [&](int j, auto band_tag) {
    constexpr int BAND = decltype(band_tag)::value;
    constexpr int AH = RDYN_CHOLQR_AHEAD;
    d4 D[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) D[cb] = (d4){0.0, 0.0, 0.0, 0.0};
    {
      // rows (cb1, kk) of W, AH rows ahead of their MFMAs and no further (compiler barrier): left alone the scheduler hoists every
      // operand of the group and spills the accumulators.  From LDS (three-wave variant) or from global memory (four waves).
      double ring[AH + 1][NB], aring[AH + 1];
      const char* const wl = lds_raw + lane * 8;
      if constexpr (WGLOBAL) asm volatile("" : "+s"(wgp));
      // (the tile's own operand of a row travels with the row of W: a lone wave has nobody to hide the LDS round trip behind)
      auto load_row = [&](int r, double (&dst)[NB], double& adst) {  // r = flat row index from the band's first row
        const int c1 = BAND + (r >> 2), k4 = r & 3;
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
      for (int cb1 = BAND; cb1 < NB; ++cb1)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
        {
          const int r = (cb1 - BAND) * 4 + kk;
          load_row(r + AH, ring[(r + AH) % (AH + 1)], aring[(r + AH) % (AH + 1)]);
          asm volatile("" ::: "memory");
          const double a = aring[r % (AH + 1)];
#pragma unroll
          for (int cb2 = cb1; cb2 < NB; ++cb2) D[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ring[r % (AH + 1)][cb2], D[cb2], 0, 0, 0);
        }
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
          if (rb >= BAND) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[rb][t], D[cb][t], acc[ti], 0, 0, 0);
          ++ti;
        }
    }
  };
  if (t_first < n_tiles) fetch(t_first);
  for (int64_t tl = t_first; tl < n_tiles; tl += t_step)
  {
    // ================================================================ sweep: my 16 samples x 4 lanes -> my tile
    {
      const bool valid = tl * 16 + s_loc < fa.n_samples;
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
#undef DUO_BARRIER
#define DUO_BARRIER()  // nobody else reads my tile
#undef RDYN_DUO_TILE_PAD
#define RDYN_DUO_TILE_PAD RDYN_SOLO_PAD
#pragma unroll
      for (int f = 0; f < NJ; ++f)
      {
#include "rdyn_duo_link_body.inc"
      }
#undef RDYN_DUO_TILE_PAD
#define RDYN_DUO_TILE_PAD 4
#undef DUO_BARRIER
#define DUO_BARRIER() asm volatile("s_barrier" ::: "memory")
      {
#include "rdyn_duo_comp_cols.inc"
      }
      {
        char* const lb = tile + fa.lds_off_b + s_loc * 8;
        if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
        if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ================================================================ consume: every row group x W, the Gram of the product
#pragma unroll
    for (int f = 0; f < NJ; ++f)
    {
      switch ((SH + 10 * f) >> 4)  // wave-uniform
      {
      case 0: row_group(f, std::integral_constant<int, 0>()); break;
      case 1: row_group(f, std::integral_constant<int, 1>()); break;
      case 2: row_group(f, std::integral_constant<int, 2>()); break;
      case 3: row_group(f, std::integral_constant<int, 3>()); break;
      case 4: row_group(f, std::integral_constant<int, 4>()); break;
      default: row_group(f, std::integral_constant<int, NB - 1>()); break;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of the tile are back before the next sweep overwrites it
    __builtin_amdgcn_wave_barrier();
  }
  // ---- block reduction (fixed order); the reduction area overlays W / the tiles
  __syncthreads();
  {
    double* const red = (double*)lds_raw;
    for (int w = 0; w < NW; ++w)
    {
      if (wave == w)
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
      __syncthreads();
    }
    double* const slab = fa.slabs + (int64_t)blockIdx.x * (NT * 256);
    for (int i = threadIdx.x; i < NT * 256; i += 64 * NW) slab[i] = red[i];
  }
}

template <int NJ, bool ALLREV, int NW, bool WGLOBAL, int SHC>
hipError_t launch_pgram_solo(const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)k_regressor_pgram_solo<NJ, ALLREV, NW, WGLOBAL, SHC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr.fetch_or(bit, std::memory_order_release);
  }
  constexpr int NB = (10 * NJ + 1 + 15) / 16 + 1, NT = NB * (NB + 1) / 2;
  size_t lds = (WGLOBAL ? 0 : (size_t)NT * 2048) + NW * (size_t)a.tile_bytes;
  if (lds < (size_t)NT * 2048) lds = (size_t)NT * 2048;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_regressor_pgram_solo<NJ, ALLREV, NW, WGLOBAL, SHC>), dim3(blocks), dim3(64 * NW), lds, st, a, W, run_flag);
  return hipGetLastError();
}
}  // namespace

// the padding of the last block in front of the natural column order, quantised DOWN to the shifts the all-revolute kernel is
// instantiated for (11: up to 14 component columns = seven first-order friction models or springs; 4: up to 21 = seven second-order
// ones; 0): a smaller shift only skips fewer zero blocks
int rdyn_cholqr_solo_col_shift(int n_joints, int n_comp_cols)
{
  const int room = 16 * ((10 * n_joints + 1 + 15) / 16 + 1) - (10 * n_joints + n_comp_cols + 1);
  return room >= 11 ? 11 : (room >= 4 ? 4 : 0);
}

// pairs: -1 = four waves, W from global memory; 1 = (A/B builds) three waves, W in LDS beside their tiles.  a.col_shift as above.
hipError_t rdyn_launch_regressor_pgram_solo(const RdynLdsGramArgs& a, const double* W, const int* run_flag, int blocks, int pairs, hipStream_t st)
{
  constexpr int NJ = 7;
  if (a.n_active != NJ) return hipErrorInvalidValue;
#ifdef RDYN_CHOLQR_SOLO3
  if (pairs == 1)
    return a.all_revolute ? launch_pgram_solo<NJ, true, 3, false, -1>(a, W, run_flag, blocks, st) : launch_pgram_solo<NJ, false, 3, false, -1>(a, W, run_flag, blocks, st);
#endif
  if (pairs != -1) return hipErrorInvalidValue;
  if (a.all_revolute)
  {
    if (a.col_shift == 11) return launch_pgram_solo<NJ, true, 4, true, 11>(a, W, run_flag, blocks, st);
    if (a.col_shift == 4) return launch_pgram_solo<NJ, true, 4, true, 4>(a, W, run_flag, blocks, st);
    if (a.col_shift == 0) return launch_pgram_solo<NJ, true, 4, true, 0>(a, W, run_flag, blocks, st);
  }
  return a.all_revolute ? launch_pgram_solo<NJ, true, 4, true, -1>(a, W, run_flag, blocks, st) : launch_pgram_solo<NJ, false, 4, true, -1>(a, W, run_flag, blocks, st);
}
