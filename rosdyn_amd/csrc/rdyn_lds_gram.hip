// rdyn_lds_gram.hip -- regressor rows -> fp64-MFMA Gram with the rows staged in LDS only (no global image at all).
//
// One wave = one tile of 16 samples.  The 64 lanes are 16 samples x 4 lanes; lane k of a sample owns the regressor
// rows (input joints) 2k and 2k+1 and runs the row-pair form of the forward local-frame sweep (rdyn_rowpair.hip:
// rolled link loop, two joint-twist slots, closed-form 10-vectors per (row, link)).  As each link is finished the
// lane drops its two 10-vectors into the wave's LDS tile, stored COLUMN-major and joint-major within a column
// (row = 16 j + sample), packed: the columns of link f only keep the rows of the joints that can be non-zero there
// (block upper-triangular Y, primitives_impl.h:1341-1347), which is what lets 16 samples x (P + 1) columns fit:
// 29.6 KB per wave at n = 6 / P = 60, 38 KB at n = 7 / P = 70.  The measured torque goes into column P.
//
// When the tile is complete the same wave runs the Gram k-steps straight out of LDS with v_mfma_f64_16x16x4_f64
// (rdyn_gram_common.h): row group j (the 16 samples of joint j) is one 16-row MFMA group, lane (c, g) reads four
// consecutive rows (32 B) of column 16 cb + c; column blocks left of joint j's first non-zero column are skipped
// (144 MFMAs per tile instead of 240 at P = 60).  Accumulators stay in registers for the whole persistent kernel; the
// epilogue and k_gram_finish are those of rdyn_gram.hip.  No __syncthreads in the main loop: a wave only reads what
// it wrote itself (LDS operations of one wave execute in order).
//
// Requirements (else rdyn_regressor_gram falls back to rdyn_fused_gram.hip): 2 <= n_active <= 7, input joints in
// chain order (row prefix property), LDS tile x 4 waves <= 160 KB.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_gram_common.h"
#include "rdyn_duo_common.h"

namespace
{

// NJ = chain joints; the number of 16-column blocks follows from it (P + 1 = 10 NJ + 1 columns).  The link loop is
// unrolled (registers are not the constraint at the one wave per SIMD the LDS tiles allow): a rolled loop has ~8
// dependent scalar-load round trips per link for the chain constants, unrolled hipcc hoists them across links.
// Measured gain is small (sweep phase 566 -> 554 us per 1e6 samples): the phase is bound by the dependent fp64 chains of
// ONE wave per SIMD, not by the constants.
template <int NJ>
__global__ __launch_bounds__(256) void k_regressor_gram_lds(const RdynLdsGramArgs fa)
{
  constexpr int NB = (10 * NJ + 1 + 15) / 16;
  constexpr int NT = NB * (NB + 1) / 2;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  ChainPtr c = as_const(fa.chain);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const tile = lds_raw + (size_t)wave * fa.tile_bytes;  // this wave's private tile
  const int s_loc = lane >> 2, k = lane & 3;                 // sweep role: sample within the tile, row pair
  const int cl = lane & 15, g = lane >> 4;                   // MFMA role: column within a block, row quad
  const int n = fa.n_active;
  constexpr int P = 10 * NJ;
  const int r0 = 2 * k, r1 = 2 * k + 1;
  // inputs of rows k and k + 4, measured torques of rows 2 k and 2 k + 1: read at the caller's input index of each (rdyn_kernels.h: in_map)
  RDYN_DUO_INPUT_OFFSETS(fa, k, in_oa, in_ob);
  const int64_t in_o0 = (int64_t)(k == 0 ? fa.in_map[0] : (k == 1 ? fa.in_map[2] : (k == 2 ? fa.in_map[4] : fa.in_map[6]))) * fa.in_sj;
  const int64_t in_o1 = (int64_t)(k == 0 ? fa.in_map[1] : (k == 1 ? fa.in_map[3] : (k == 2 ? fa.in_map[5] : fa.in_map[7]))) * fa.in_sj;

  // MFMA role: LDS byte offset of my column in every column block and the number of joint row-groups it stores
  int colbase[NB], colm[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
  {
    const int p = 16 * cb + cl;
    if (p < P)
    {
      const int f = p / 10;
      colbase[cb] = fa.lds_off[f] + (p - 10 * f) * fa.lds_stride[f];
      colm[cb] = fa.lds_m[f];
    }
    else if (p == P)
    {
      colbase[cb] = fa.lds_off_b;
      colm[cb] = n;
    }
    else
    {
      colbase[cb] = 0;
      colm[cb] = 0;
    }
  }

  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

  // Inputs: lane k of a sample's quad holds input joints k and k + 4 (q, Dq, DDq: six doubles); the link loop gets
  // its joint's values by a quad shuffle.  At one wave per SIMD a global load inside the link loop would expose the
  // full HBM latency once per link; fetched one tile ahead (behind the previous tile's MFMAs) it is hidden.
  double nqa = 0.0, ndqa = 0.0, nddqa = 0.0, nqb = 0.0, ndqb = 0.0, nddqb = 0.0;
  double nb0 = 0.0, nb1 = 0.0;  // measured torque of my two rows, fetched with the inputs
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
    // ================= sweep: my sample, my two rows, all links -> LDS tile
    const int64_t s = tl * 16 + s_loc;
    const bool valid = s < fa.n_samples;
    const double zmask = valid ? 1.0 : 0.0;
    // this tile's inputs were fetched during the previous tile's Gram phase (or in the prologue)
    const double qa = nqa, dqa = ndqa, ddqa = nddqa, qb = nqb, dqb = ndqb, ddqb = nddqb;
    const double tb0 = nb0 * zmask, tb1 = nb1 * zmask;

    V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
    V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
    V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);

    const bool skip_sweep = (fa.debug & 1) && tl != t_first;  // debug bit 0: sweep only the first tile (timing)
    if (!skip_sweep)
#pragma unroll(NJ <= 7 ? NJ : 1)  // longer chains: rolled (unrolled, NJ = 10 needs scratch)
    for (int f = 0; f < NJ; ++f)
    {
      JointRef J = c->j[f];
      const int type = J.type;
      const int idx = J.in_idx;
      double qf = 0.0, dqf = 0.0, ddqf = 0.0;
      if (idx >= 0)
      {
        // input joint idx lives in lane (idx & 3) of my sample's quad, first or second slot (wave-uniform choice)
        const int src = (lane & ~3) | (idx & 3);
        const bool second = idx >= 4;
        qf = __shfl(second ? qb : qa, src);
        dqf = __shfl(second ? dqb : dqa, src);
        ddqf = __shfl(second ? ddqb : ddqa, src);
      }
      double R[9];
      V3 tt = ld3(J.t);
      if (type == RDYN_REVOLUTE)
      {
        double sn, cs;
        rdyn_sincos(qf, &sn, &cs);
        const double oc = 1.0 - cs;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = fma(sn, J.B[i], fma(oc, J.C[i], J.A[i]));
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = J.A[i];
        if (type == RDYN_PRISMATIC) tt = axpy(tt, ld3(J.up), qf);
      }
      {
        const V3 wn = rotT(R, w);
        const V3 vn = rotT(R, vl + cross(w, tt));
        const V3 aln = rotT(R, al);
        const V3 an = rotT(R, lin + cross(al, tt));
        w = wn; vl = vn; al = aln; lin = an;
        const V3 nL0 = rotT(R, L0 + cross(A0, tt));
        A0 = rotT(R, A0);
        L0 = nL0;
        const V3 nL1 = rotT(R, L1 + cross(A1, tt));
        A1 = rotT(R, A1);
        L1 = nL1;
      }
      const V3 u = ld3(J.u);
      V3 sl = mk(0, 0, 0), sa = mk(0, 0, 0);
      if (type == RDYN_REVOLUTE)
      {
        lin = axpy(lin, cross(vl, u), dqf);
        al = axpy(axpy(al, cross(w, u), dqf), u, ddqf);
        w = axpy(w, u, dqf);
        sa = u;
      }
      else if (type == RDYN_PRISMATIC)
      {
        lin = axpy(axpy(lin, cross(w, u), dqf), u, ddqf);
        vl = axpy(vl, u, dqf);
        sl = u;
      }
      if (idx >= 0)
      {
        const bool m0 = (idx == r0), m1 = (idx == r1);
        L0 = mk(m0 ? sl.x : L0.x, m0 ? sl.y : L0.y, m0 ? sl.z : L0.z);
        A0 = mk(m0 ? sa.x : A0.x, m0 ? sa.y : A0.y, m0 ? sa.z : A0.z);
        L1 = mk(m1 ? sl.x : L1.x, m1 ? sl.y : L1.y, m1 ? sl.z : L1.z);
        A1 = mk(m1 ? sa.x : A1.x, m1 ? sa.y : A1.y, m1 ? sa.z : A1.z);
      }

      const V3 d = lin + cross(w, vl);
      const double wxy = w.x * w.y, wxz = w.x * w.z, wyz = w.y * w.z;
      const double wxx = w.x * w.x, wyy = w.y * w.y, wzz = w.z * w.z;
      const double b00 = -(wyy + wzz), b01 = wxy - al.z, b02 = wxz + al.y;
      const double b10 = wxy + al.z, b11 = -(wxx + wzz), b12 = wyz - al.x;
      const double b20 = wxz - al.y, b21 = wyz + al.x, b22 = -(wxx + wyy);
      double y0[10], y1[10];
      {
        const V3 dxA = cross(d, A0), x = cross(A0, w);
        y0[0] = dot(L0, d);
        y0[1] = fma(L0.x, b00, fma(L0.y, b10, fma(L0.z, b20, dxA.x)));
        y0[2] = fma(L0.x, b01, fma(L0.y, b11, fma(L0.z, b21, dxA.y)));
        y0[3] = fma(L0.x, b02, fma(L0.y, b12, fma(L0.z, b22, dxA.z)));
        y0[4] = fma(A0.x, al.x, x.x * w.x);
        y0[5] = fma(A0.x, al.y, fma(A0.y, al.x, fma(x.x, w.y, x.y * w.x)));
        y0[6] = fma(A0.x, al.z, fma(A0.z, al.x, fma(x.x, w.z, x.z * w.x)));
        y0[7] = fma(A0.y, al.y, x.y * w.y);
        y0[8] = fma(A0.y, al.z, fma(A0.z, al.y, fma(x.y, w.z, x.z * w.y)));
        y0[9] = fma(A0.z, al.z, x.z * w.z);
      }
      {
        const V3 dxA = cross(d, A1), x = cross(A1, w);
        y1[0] = dot(L1, d);
        y1[1] = fma(L1.x, b00, fma(L1.y, b10, fma(L1.z, b20, dxA.x)));
        y1[2] = fma(L1.x, b01, fma(L1.y, b11, fma(L1.z, b21, dxA.y)));
        y1[3] = fma(L1.x, b02, fma(L1.y, b12, fma(L1.z, b22, dxA.z)));
        y1[4] = fma(A1.x, al.x, x.x * w.x);
        y1[5] = fma(A1.x, al.y, fma(A1.y, al.x, fma(x.x, w.y, x.y * w.x)));
        y1[6] = fma(A1.x, al.z, fma(A1.z, al.x, fma(x.x, w.z, x.z * w.x)));
        y1[7] = fma(A1.y, al.y, x.y * w.y);
        y1[8] = fma(A1.y, al.z, fma(A1.z, al.y, fma(x.y, w.z, x.z * w.y)));
        y1[9] = fma(A1.z, al.z, x.z * w.z);
      }
      // rows j < m_f are stored for the columns of link f (row = 16 j + sample)
      const int mf = fa.lds_m[f], stride = fa.lds_stride[f];
      char* const lf = tile + fa.lds_off[f] + s_loc * 8;
      if (r0 < mf)
      {
#pragma unroll
        for (int p = 0; p < 10; ++p) *(double*)(lf + p * stride + r0 * 128) = y0[p] * zmask;
      }
      if (r1 < mf)
      {
#pragma unroll
        for (int p = 0; p < 10; ++p) *(double*)(lf + p * stride + r1 * 128) = y1[p] * zmask;
      }
    }
    // measured torque -> column P
    {
      char* const lb = tile + fa.lds_off_b + s_loc * 8;
      if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
      if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (tl + t_step < n_tiles) fetch(tl + t_step);  // next tile's inputs travel behind this tile's MFMAs

    // ================= Gram: row group j = the 16 samples of input joint j
    // (operands of group j + 1 are read from LDS while the MFMAs of group j run)
    auto lds_load = [&](int j, int cbm, d4* v) {
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
      {
        d4 x = (d4){0.0, 0.0, 0.0, 0.0};
        if (cb >= cbm && j < colm[cb]) x = *(const d4*)(tile + colbase[cb] + j * 128 + g * 32);
        v[cb] = x;
      }
    };
    const int n_groups = (fa.debug & 2) ? 0 : n;  // debug bit 1: no Gram phase (timing)
    // double-buffered operands only where the registers are there: with NB >= 5 (15+ accumulator tiles) the second
    // operand set pushed the kernel into scratch (156 B, 6.7 ms vs 5.4 ms at n = 7)
    constexpr bool PF = NB <= 4;
    d4 cur[NB], nxt[PF ? NB : 1];
    int cbm = 0, cbm_n = 0;
    if (PF && n_groups > 0)
    {
      cbm = fa.first_col[0] >> 4;
      lds_load(0, cbm, cur);
    }
#pragma nounroll
    for (int j = 0; j < n_groups; ++j)
    {
      if (PF)
      {
        if (j + 1 < n_groups)
        {
          cbm_n = fa.first_col[j + 1] >> 4;
          lds_load(j + 1, cbm_n, nxt);
        }
      }
      else
      {
        cbm = fa.first_col[j] >> 4;
        lds_load(j, cbm, cur);
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
      if (PF)
      {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) cur[cb] = nxt[PF ? cb : 0];
        cbm = cbm_n;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next sweep
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // ================= epilogue: block reduction through LDS (the tiles are dead now), this block's Gram slab
  __syncthreads();
  double* red = (double*)lds_raw;
  gram_block_reduce_to_slab<NT>(acc, red, wave, cl, g, fa.slabs + (int64_t)blockIdx.x * (NT * 256), false);
}

template <int NJ>
hipError_t launch_lds_nj(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  // > 64 KB of dynamic LDS needs the opt-in attribute, once per instantiation AND device (one bit per device ordinal)
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)k_regressor_gram_lds<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_regressor_gram_lds<NJ>), dim3(blocks), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}
}  // namespace

hipError_t rdyn_launch_regressor_gram_lds(int n_cols, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, hipStream_t st)
{
  switch (n_cols / 10)  // chain joints
  {
  case 2: return launch_lds_nj<2>(a, blocks, lds_bytes, st);
  case 3: return launch_lds_nj<3>(a, blocks, lds_bytes, st);
  case 4: return launch_lds_nj<4>(a, blocks, lds_bytes, st);
  case 5: return launch_lds_nj<5>(a, blocks, lds_bytes, st);
  case 6: return launch_lds_nj<6>(a, blocks, lds_bytes, st);
  case 7: return launch_lds_nj<7>(a, blocks, lds_bytes, st);
  case 8: return launch_lds_nj<8>(a, blocks, lds_bytes, st);
  case 9: return launch_lds_nj<9>(a, blocks, lds_bytes, st);
  case 10: return launch_lds_nj<10>(a, blocks, lds_bytes, st);
  default: return hipErrorInvalidValue;
  }
}
