// rdyn_pipe_gram.hip -- the LDS-tile regressor -> Gram kernel (rdyn_lds_gram.hip) software-pipelined INSIDE the wave.
//
// rdyn_lds_gram.hip runs two phases per 16-sample tile, one after the other on the same wave: the fp64 VALU sweep that
// fills the LDS tile, then the fp64 MFMA k-steps that consume it.  LDS capacity (one 30 KB tile per wave, four waves per
// CU) rules out a second wave per SIMD, so each phase runs with the dependency stalls of a single wave: sweep 0.55 ms +
// Gram 0.47 ms per 1e6 samples at n = 6.  Here the two phases share ONE instruction stream:
//   * row group j of tile t (the 16 samples of input joint j) is consumed DURING THE SWEEP OF TILE t + 1, at link j: its
//     rows only exist in the columns of links >= j, which that sweep has not rewritten yet (link j's own columns are
//     written at the end of link j's block), so the operands are read from LDS just in time -- one group (<= NB quads)
//     in registers at a time, zero band skipped at compile time ((10 j) >> 4, input joints in chain order);
//   * __builtin_amdgcn_sched_group_barrier interleaves the group's MFMAs with the VALU stream of the link, one
//     v_mfma_f64_16x16x4_f64 per RDYN_PIPE_VALU_PER_MFMA VALU instructions, so each fills the other's dependency stalls;
//   * to give the scheduler one large straight-line region per link the sweep is written branch-free after the sincos:
//     joint kinds are handled by selects / zero factors (a non-revolute joint takes rdyn_sincos(0): R = A exactly), rows that
//     are not stored for a link are written to a per-lane dummy slot instead of being branched around;
//   * the chain pointer is laundered once per tile: hoisted out of the tile loop the unrolled links' constants need ~450
//     SGPRs and come back as v_readlane traffic.
// Measured (same box, n = 6 / P = 60 / N = 1e6): 0.82-0.86 ms vs 0.98-1.01 ms for the two-phase kernel (+18 %).
// What it cannot do: run the matrix pipe BEHIND the VALU.  On gfx950 fp64 MFMA and fp64 VALU do not overlap
// (tools/mfma_valu_overlap.hip: MFMA-only 5.10 ms, FMA-only 2.63 ms, interleaved 7.49 ms = the sum), and the ratio of
// VALU instructions per MFMA makes no difference (5, 10, 16: same time): the gain is stall filling, the floor is the SUM
// of the two instruction streams (~18 k cycles per tile; the kernel runs at ~31 k, PMC in profiles/r1/pipe_gram_pmc.txt).
// Same tile layout, arguments, accumulation order and epilogue as rdyn_lds_gram.hip: results are bit-identical to it.
// Chains of up to 6 joints: link loop unrolled.  7 joints (5 column blocks, 15 accumulator tiles): unrolled, hipcc spills
// 352 B and the kernel is slower than the two-phase one (6.33 vs 5.14 ms at n = 7, N = 4e6); with the link loop ROLLED
// and the body instantiated once per zero band it spills 44 B and wins (4.90 ms).  8+ joints: more spills, no gain
// (2.97 vs 2.92 ms at 9 joints) -- those chains keep the two-phase kernel.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_gram_common.h"
#include "rdyn_duo_common.h"

#ifndef RDYN_PIPE_UNROLL_MAX
#define RDYN_PIPE_UNROLL_MAX 6  // chains up to this many joints: unrolled link loop; longer: rolled (A/B: tools/probe_pipe.py)
#endif
#ifndef RDYN_PIPE_MFMA_BURST
#define RDYN_PIPE_MFMA_BURST 1  // MFMAs issued back to back before the VALU group (an MFMA <-> fp64 VALU switch costs ~80 cycles: tools/fp64_issue.hip)
#endif
#ifndef RDYN_PIPE_VALU_PER_MFMA
#define RDYN_PIPE_VALU_PER_MFMA 10  // VALU instructions scheduled behind every MFMA (A/B: tools/probe_pipe.py)
#endif

namespace
{

// one k-step (element t of the operand quads) of one 16-row group: all upper tiles with both column blocks >= cbm
template <int NB>
__device__ __forceinline__ void mfma_kstep(const d4* op, int t, int cbm, d4* acc)
{
  int ti = 0;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
#pragma unroll
    for (int rb = 0; rb <= cb; ++rb)
    {
      if (rb >= cbm) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[rb][t], op[cb][t], acc[ti], 0, 0, 0);
      ++ti;
    }
}
__device__ __forceinline__ constexpr int tiles_from(int nb, int cbm) { return (nb - cbm) * (nb - cbm + 1) / 2; }
template <int NJ, bool ROLLED>
__global__ __launch_bounds__(256) void k_regressor_gram_pipe(const RdynLdsGramArgs fa)
{
  constexpr int NB = (10 * NJ + 1 + 15) / 16;
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int P = 10 * NJ;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  ChainPtr c = as_const(fa.chain);  // re-laundered per tile, see below
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const tile = lds_raw + (size_t)wave * fa.tile_bytes;  // this wave's private tile
  char* const dummy = tile + fa.lds_dummy_off + lane * 8;     // where rows that a link does not store are dropped
  const int s_loc = lane >> 2, k = lane & 3;                 // sweep role: sample within the tile, row pair
  const int cl = lane & 15, g = lane >> 4;                   // MFMA role: column within a block, row quad
  const int n = fa.n_active;
  const int r0 = 2 * k, r1 = 2 * k + 1;
  // inputs of rows k and k + 4, measured torques of rows 2 k and 2 k + 1: read at the caller's input index of each (rdyn_kernels.h: in_map)
  RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
  const int64_t in_o0 = (int64_t)(k == 0 ? fa.in_map[0] : (k == 1 ? fa.in_map[2] : (k == 2 ? fa.in_map[4] : fa.in_map[6]))) * fa.in_sj;
  const int64_t in_o1 = (int64_t)(k == 0 ? fa.in_map[1] : (k == 1 ? fa.in_map[3] : (k == 2 ? fa.in_map[5] : fa.in_map[7]))) * fa.in_sj;

  int colbase[NB], colm[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
  {
    const int p = 16 * cb + cl;
    const int f = p < P ? p / 10 : 0;
    colbase[cb] = p < P ? fa.lds_off[f] + (p - 10 * f) * fa.lds_stride[f] : (p == P ? fa.lds_off_b : 0);
    colm[cb] = p < P ? fa.lds_m[f] : (p == P ? n : 0);
  }

  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  // operands of row group j (the 16 samples of input joint j) out of this wave's tile: column blocks >= (10 j) >> 4
  auto lds_group = [&](int j, d4* op) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
    {
      d4 x = (d4){0.0, 0.0, 0.0, 0.0};
      if (cb >= ((10 * j) >> 4) && j < colm[cb]) x = *(const d4*)(tile + colbase[cb] + j * 128 + g * 32);
      op[cb] = x;
    }
  };
  // before the first tile the "previous tile" is all zeros
  for (int i = lane * 8; i < fa.tile_bytes; i += 64 * 8) *(double*)(tile + i) = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // inputs one tile ahead, as in rdyn_lds_gram.hip: lane k of a sample's quad holds input joints k and k + 4
  double nqa = 0.0, ndqa = 0.0, nddqa = 0.0, nqb = 0.0, ndqb = 0.0, nddqb = 0.0, nb0 = 0.0, nb1 = 0.0;
  auto fetch = [&](int64_t tile_index) {
    int64_t sx = tile_index * 16 + s_loc;
    if (sx >= fa.n_samples) sx = fa.n_samples - 1;
    const int64_t o = sx * fa.in_ss;
    if (fa.bcol)
    {
      if (r0 < n) nb0 = fa.bcol[o + in_o0];
      if (r1 < n) nb1 = fa.bcol[o + in_o1];
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
  const int64_t n_tiles = (fa.n_samples + 15) / 16;
  const int64_t t_first = (int64_t)blockIdx.x * 4 + wave, t_step = (int64_t)gridDim.x * 4;
  if (t_first < n_tiles) fetch(t_first);
  for (int64_t tl = t_first; tl < n_tiles; tl += t_step)
  {
    // keep the chain constants' scalar loads inside the iteration: hoisted out of the tile loop they need ~450 SGPRs,
    // which end up as v_writelane / v_readlane traffic in the VALU stream
    asm volatile("" : "+s"(c));
    const double zmask = (tl * 16 + s_loc < fa.n_samples) ? 1.0 : 0.0;
    const double qa = nqa, dqa = ndqa, ddqa = nddqa, qb = nqb, dqb = ndqb, ddqb = nddqb;
    const double tb0 = nb0 * zmask, tb1 = nb1 * zmask;

    V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
    V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
    V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);

    if constexpr (!ROLLED)
    {
#pragma unroll
      for (int f = 0; f < NJ; ++f)
      {
        const int CBM = (10 * f) >> 4;
#include "rdyn_pipe_link_body.inc"
      }
    }
    else
    {
      // longer chains: the unrolled form needs more registers than the file has (15+ accumulator tiles), so the link loop
      // stays rolled and the body is instantiated once per zero band (CBM selects the MFMAs of row group f at compile time)
#pragma nounroll
      for (int f = 0; f < NJ; ++f)
      {
        switch ((10 * f) >> 4)
        {
        case 0:
        {
          constexpr int CBM = 0;
#include "rdyn_pipe_link_body.inc"
        }
        break;
        case 1:
          if constexpr (1 < NB)
          {
            constexpr int CBM = 1;
#include "rdyn_pipe_link_body.inc"
          }
          break;
        case 2:
          if constexpr (2 < NB)
          {
            constexpr int CBM = 2;
#include "rdyn_pipe_link_body.inc"
          }
          break;
        case 3:
          if constexpr (3 < NB)
          {
            constexpr int CBM = 3;
#include "rdyn_pipe_link_body.inc"
          }
          break;
        case 4:
          if constexpr (4 < NB)
          {
            constexpr int CBM = 4;
#include "rdyn_pipe_link_body.inc"
          }
          break;
        default:
          if constexpr (5 < NB)
          {
            constexpr int CBM = 5;
#include "rdyn_pipe_link_body.inc"
          }
          break;
        }
      }
    }
    // measured torque -> column P
    {
      char* const lb = tile + fa.lds_off_b + s_loc * 8;
      *(double*)(r0 < n ? lb + r0 * 128 : dummy) = tb0;
      *(double*)(r1 < n ? lb + r1 * 128 : dummy) = tb1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (tl + t_step < n_tiles) fetch(tl + t_step);

  }
  // the last tile's row groups (all zero if this wave had no tile)
#pragma unroll
  for (int j = 0; j < NJ; ++j)
  {
    d4 op[NB];
    lds_group(j, op);
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma_kstep<NB>(op, t, (10 * j) >> 4, acc);  // operands left of the band are zero
  }

  // ================= epilogue: block reduction through LDS (the tiles are dead now), this block's Gram slab
  __syncthreads();
  double* red = (double*)lds_raw;
  gram_block_reduce_to_slab<NT>(acc, red, wave, cl, g, fa.slabs + (int64_t)blockIdx.x * (NT * 256), false);
}

template <int NJ>
hipError_t launch_pipe_nj(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  // > 64 KB of dynamic LDS needs the opt-in attribute, once per instantiation AND device (one bit per device ordinal)
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)k_regressor_gram_pipe<NJ, (NJ > RDYN_PIPE_UNROLL_MAX)>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_regressor_gram_pipe<NJ, (NJ > RDYN_PIPE_UNROLL_MAX)>), dim3(blocks), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}
}  // namespace

bool rdyn_regressor_gram_pipe_supported(int n_cols) { return n_cols >= 20 && n_cols <= 70; }

hipError_t rdyn_launch_regressor_gram_pipe(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  switch (n_cols / 10)  // chain joints
  {
  case 2: return launch_pipe_nj<2>(a, blocks, lds_bytes, st);
  case 3: return launch_pipe_nj<3>(a, blocks, lds_bytes, st);
  case 4: return launch_pipe_nj<4>(a, blocks, lds_bytes, st);
  case 5: return launch_pipe_nj<5>(a, blocks, lds_bytes, st);
  case 6: return launch_pipe_nj<6>(a, blocks, lds_bytes, st);
  case 7: return launch_pipe_nj<7>(a, blocks, lds_bytes, st);
  default: return hipErrorInvalidValue;
  }
}
