// rdyn_fused_gram.hip -- regressor rows -> fp64-MFMA Gram in ONE persistent kernel: the regressor image never
// makes a round trip through HBM.
//
// Every workgroup (256 threads, one per CU) walks tiles of 256 samples:
//   phase 1  the forward local-frame sweep of rdyn_kernels.hip (same included body, MODE_REGRESSOR_GRAM, one
//            thread per sample) writes the tile's element-major regressor image (rows j*256 + s, columns 0..P with
//            the measured torque in column P) into THIS WORKGROUP'S image in the workspace.  The image is rewritten
//            for every tile, so it lives in L2 / the 256 MiB Infinity Cache, not in HBM;
//   barrier
//   phase 2  the four waves run the Gram k-steps of rdyn_gram.hip over the tile's n*256 rows (v_mfma_f64_16x16x4_f64,
//            structure-aware, three 16-row groups in flight) into accumulators that stay in registers for the whole kernel;
//   barrier  (the next tile overwrites the image).
// At the end the accumulators go through LDS into the workgroup's Gram slab and k_gram_finish sums the slabs in
// fixed order, exactly as in the unfused path.  HBM traffic: the 4 n doubles of input per sample.
//
// Measured (MI355X, 1e6 samples, n = 6, P = 60): 1.10 ms, phase 1 alone 0.46 ms, phase 2 alone 0.58 ms -- both
// latency-bound at the one wave per SIMD the 460-register footprint allows, and serialised.  Tried and rejected:
//   * two workgroups per CU via __launch_bounds__(256, 2): 388 B of scratch spills, 1.25-1.36 ms;
//   * wave specialisation (4 sweep waves + 4 Gram waves per workgroup, double-buffered images, VALU || MFMA on every
//     SIMD): the kernel-wide register allocation must cover both roles inside 256 registers -> 444 B of spills,
//     1.49 ms (n = 7: 20 ms).  It needs per-role register budgets, i.e. two cooperating kernels.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_record_stage.h"
#include "rdyn_gram_common.h"

namespace
{
enum
{
  MODE_REGRESSOR = 0,
  MODE_TORQUE = 1,
  MODE_INERTIA = 2,
  MODE_REGRESSOR_GRAM = 3,
  MODE_REGRESSOR_EXPAND = 4,
  MODE_REGRESSOR_EXPAND_STAGED = 5
};
#define RDYN_IS_EXPAND(MODE) ((MODE) == MODE_REGRESSOR_EXPAND || (MODE) == MODE_REGRESSOR_EXPAND_STAGED)
#define RDYN_IS_REGRESSOR(MODE) ((MODE) == MODE_REGRESSOR || (MODE) == MODE_REGRESSOR_GRAM || RDYN_IS_EXPAND(MODE))
#define RDYN_BODY_EXIT break

template <int NJ, int NB>
__global__ __launch_bounds__(256) void k_regressor_gram_fused(const RdynFusedGramArgs fa)
{
  constexpr int NT = NB * (NB + 1) / 2;
  constexpr int MODE = MODE_REGRESSOR_GRAM;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cl = lane & 15, g = lane >> 4;
  const int n = fa.n_active, P = 10 * NJ;
  const int64_t lda = (int64_t)n * 256;                          // rows of one tile image
  double* const img = fa.images + (int64_t)blockIdx.x * lda * (P + 1);

  // per-lane column base pointers into this workgroup's image (padding columns -> null)
  const double* col[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
  {
    const int p = 16 * cb + cl;
    col[cb] = (p <= P) ? img + (int64_t)p * lda : nullptr;
  }
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

  const int64_t n_tiles = (fa.sweep.n_samples + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
  {
    // ---------------- phase 1: sweep of this tile's samples into the image
    if (!(fa.debug & 1) || tile == (int64_t)blockIdx.x)  // debug bit 0: timing-only run of phase 2 (image written once)
    {
      RdynSweepArgs a = fa.sweep;
      const int64_t s0 = tile * 256;
      a.q += s0 * a.in_ss;
      a.dq += s0 * a.in_ss;
      a.ddq += s0 * a.in_ss;
      if (a.bcol) a.bcol += s0 * a.in_ss;
      a.n_samples = fa.sweep.n_samples - s0 < 256 ? fa.sweep.n_samples - s0 : 256;
      a.tau = nullptr;
      a.Y = img;
      a.y_ss = 1;
      a.y_sr = 256;
      a.y_sc = lda;
      const unsigned blk = 0;
      constexpr double* expand_tile = nullptr;  // (rdyn_kernels.hip: k_expand_staged)
      bool done = false;
      do
      {
#include "rdyn_local_sweep_body.inc"
        done = true;
      } while (0);
      if (!done)
      {
        // lanes beyond the batch (last tile only): their rows of the image must read as zero
        ChainPtr c = as_const(a.chain);
        for (int l = 0; l < NJ; ++l)
        {
          const int r = c->j[l].in_idx;
          if (r < 0) continue;
          for (int p = ((10 * l) / 16) * 16; p <= P; ++p) img[(int64_t)p * lda + r * 256 + threadIdx.x] = 0.0;
        }
      }
    }
    __syncthreads();

    // ---------------- phase 2: Gram k-steps over the tile's n * 256 rows (16-row groups, 4 waves)
    if (!(fa.debug & 2))  // debug bit 1: timing-only run of phase 1
    {
      const int n_groups = n * 16;
      // three 16-row groups in flight per wave (the image is read from L2 / Infinity Cache at one wave per SIMD:
      // a single prefetched group leaves the MFMAs waiting on the loads)
      d4 cur[NB], nxt[NB], n2[NB];
      auto cbm_of = [&](int grp) -> int { return fa.first_col[grp >> 4] >> 4; };  // 16 groups per row block of 256
      auto load = [&](int grp, int cbm, d4* v) {
        const int64_t r = (int64_t)grp * 16 + 4 * g;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
        {
          d4 x = (d4){0.0, 0.0, 0.0, 0.0};
          if (cb >= cbm && col[cb]) x = *(const d4*)(col[cb] + r);
          v[cb] = x;
        }
      };
      int grp = wave;
      int cbm = 0, cbm_1 = 0, cbm_2 = 0;
      if (grp < n_groups)
      {
        cbm = cbm_of(grp);
        load(grp, cbm, cur);
      }
      if (grp + 4 < n_groups)
      {
        cbm_1 = cbm_of(grp + 4);
        load(grp + 4, cbm_1, nxt);
      }
      while (grp < n_groups)
      {
        if (grp + 8 < n_groups)
        {
          cbm_2 = cbm_of(grp + 8);
          load(grp + 8, cbm_2, n2);
        }
        switch (NB > 1 ? cbm : 0)
        {
        case 0: mfma_group<NB, 0>(cur, acc); break;
        case 1: mfma_group<NB, 1>(cur, acc); break;
        case 2: mfma_group<NB, 2>(cur, acc); break;
        case 3: mfma_group<NB, 3>(cur, acc); break;
        case 4: mfma_group<NB, 4>(cur, acc); break;
        case 5: mfma_group<NB, 5>(cur, acc); break;
        default: mfma_group<NB, 6>(cur, acc); break;
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
        {
          cur[cb] = nxt[cb];
          nxt[cb] = n2[cb];
        }
        cbm = cbm_1;
        cbm_1 = cbm_2;
        grp += 4;
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue: block reduction in LDS, this block's Gram slab
  __shared__ double red[NT * 256];
  gram_block_reduce_to_slab<NT>(acc, red, wave, cl, g, fa.slabs + (int64_t)blockIdx.x * (NT * 256), false);
}

template <int NJ>
hipError_t launch_fused_nj(const RdynFusedGramArgs& a, int blocks, hipStream_t st)
{
  constexpr int NB = (10 * NJ + 1 + 15) / 16;
  hipLaunchKernelGGL((k_regressor_gram_fused<NJ, NB>), dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}
}  // namespace

hipError_t rdyn_launch_regressor_gram_fused(int n_joints, const RdynFusedGramArgs& a, int blocks, hipStream_t st)
{
  switch (n_joints)
  {
  case 1: return launch_fused_nj<1>(a, blocks, st);
  case 2: return launch_fused_nj<2>(a, blocks, st);
  case 3: return launch_fused_nj<3>(a, blocks, st);
  case 4: return launch_fused_nj<4>(a, blocks, st);
  case 5: return launch_fused_nj<5>(a, blocks, st);
  case 6: return launch_fused_nj<6>(a, blocks, st);
  case 7: return launch_fused_nj<7>(a, blocks, st);
  case 8: return launch_fused_nj<8>(a, blocks, st);
  case 9: return launch_fused_nj<9>(a, blocks, st);
  case 10: return launch_fused_nj<10>(a, blocks, st);
  default: return hipErrorInvalidValue;
  }
}
