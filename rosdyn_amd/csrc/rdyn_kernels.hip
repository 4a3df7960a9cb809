// rdyn_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): batched rigid-body dynamics sweeps.
//
// One thread per trajectory sample (wave64: 64 samples per wavefront, 256-thread workgroups).
// Chain constants are read through wave-uniform addresses -> scalar loads into SGPRs (see rdyn_device.h).
//
// k_local_sweep<NJ, MODE>  -- the north-star kernel family.  A single FORWARD sweep in link-local
//   coordinates produces, per link f (base -> tool):
//     w, vl      angular / linear velocity of the link origin           (reference: getTwist,  primitives_impl.h:1004-1009)
//     al, a      angular / linear spatial acceleration, gravity folded  (reference: getDTwist, primitives_impl.h:1113-1118)
//     j_l        the unit twist of every upstream joint l <= f, referred to link f's origin and axes
//                (= rot(R_f^T, spatialTranslation(S_l, p_f - p_l)); it is the transposed operator the
//                reference applies to the wrench regressor at primitives_impl.h:1341-1347, and the
//                link-f Jacobian column of primitives_impl.h:1371-1372)
//   and from them, without materialising any 6x10 block (and, the torque mode excepted, without a backward pass):
//     MODE_REGRESSOR  Y(l, 10 f .. 10 f + 9) = j_l^T W'_f   with the closed form of the reference's ten
//                     basis-matrix products (primitives_impl.h:1324-1339):
//                        W'_f = [ d | [al]x + [w]x[w]x | 0 ;  0 | -[d]x | L(al) + [w]x L(w) ],   d = a + w x vl
//                     and tau = Y * pi fused (== getJointTorque, primitives_impl.h:1264-1272);
//     MODE_TORQUE     tau_l = sum_f j_l . (I_f a_f + v_f x* I_f v_f + gravity)   (primitives_impl.h:1240-1257, 1270) -- evaluated as the
//                     reference does, by a backward pass over the link wrenches (S_l . sum of the downstream wrenches about joint l), the
//                     joint transforms rebuilt from the saved sin / 1 - cos: O(n) instead of n (n + 1) / 2 carried unit twists
//     MODE_INERTIA    M(l1,l2) = sum_f j_l1^T I_f j_l2                            (primitives_impl.h:1357-1379)
//   Structural zeros of the regressor (rows of joints downstream of link f) are written explicitly: the
//   output is the reference's dense n x P matrix.
//
// k_base_sweep<NJ> -- base-frame kinematics exactly as the reference states them (frames, screws, twists,
//   spatial accelerations, tool Jacobian): primitives_impl.h:863-882, 927-949, 981-1013, 1082-1124.
//
// HBM traffic per sample (fp64): regressor 3n*8 in, (n + n*P)*8 out  (3 072 B for n=6, P=60); the
// arithmetic (~3 k fp64 VALU ops per sample for n=6) is ~5x below the HBM time at 1 thread/sample, so
// the store path decides the speed: with the element-major layout every store instruction of a wave
// writes 512 contiguous bytes.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_record_stage.h"

namespace
{

enum
{
  MODE_REGRESSOR = 0,
  MODE_TORQUE = 1,
  MODE_INERTIA = 2,
  // regressor image that only the Gram kernel will read (rdyn_regressor_gram): structural zeros are stored only
  // where k_gram still loads them, i.e. inside 16-column blocks that also hold non-zero columns of that row
  MODE_REGRESSOR_GRAM = 3,
  // dense regressor of a chain LONGER than the kernels sweep: the reduced companion is swept and every (row, body) ten-vector is
  // multiplied by the constant 10 x 10 blocks of the chain links that ride on that body (Y_f = Y_body X_f, rdyn_chain.hpp)
  MODE_REGRESSOR_EXPAND = 4,
  // the same into a row-contiguous layout (per-sample images, stacked matrix): 64-thread workgroups, every link's block staged in the
  // wave's LDS tile and copied out 16 bytes per lane (8-byte pieces scattered over 64 lines per store instruction: 3-4 ms per 1e6)
  MODE_REGRESSOR_EXPAND_STAGED = 5
};
#define RDYN_IS_EXPAND(MODE) ((MODE) == MODE_REGRESSOR_EXPAND || (MODE) == MODE_REGRESSOR_EXPAND_STAGED)
#define RDYN_IS_REGRESSOR(MODE) ((MODE) == MODE_REGRESSOR || (MODE) == MODE_REGRESSOR_GRAM || RDYN_IS_EXPAND(MODE))
#define RDYN_BODY_EXIT return

// one chain, one batch: grid.x = ceil(N / 256)
#ifdef RDYN_LOCAL_WAVES  // A/B builds only: an occupancy target for the one-thread-per-sample sweeps
#define RDYN_LOCAL_ATTR __attribute__((amdgpu_waves_per_eu(RDYN_LOCAL_WAVES, RDYN_LOCAL_WAVES)))
#else
#define RDYN_LOCAL_ATTR
#endif
template <int NJ, int MODE>
__global__ __launch_bounds__(256) RDYN_LOCAL_ATTR void k_local_sweep(const RdynSweepArgs a)
{
  const unsigned blk = blockIdx.x;
  constexpr double* expand_tile = nullptr;  // (the staged expanding sweep's LDS tile: k_expand_staged)
#include "rdyn_local_sweep_body.inc"
}
// torque / inertia in the sample-major layout (a.staged = doubles per record): one wave per workgroup, the wave's 64 records through its
// LDS tile (64 (rec | 1) doubles of dynamic LDS), written in whole lines
template <int NJ, int MODE>
__global__ __launch_bounds__(64) RDYN_LOCAL_ATTR void k_local_sweep_rec(const RdynSweepArgs a)
{
  const unsigned blk = blockIdx.x;
  constexpr double* expand_tile = nullptr;
#define RDYN_LOCAL_WGS 64
#include "rdyn_local_sweep_body.inc"
}
// the staged expanding sweep: one wave per workgroup, its tile in dynamic LDS (64 x (CG NJ + 2) doubles, CG columns of a link at a time)
// and behind it the lanes' inputs (64 x (3 NJ + 1) doubles)
#ifndef RDYN_EXPAND_WAVES
#define RDYN_EXPAND_WAVES 2
#endif
template <int NJ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RDYN_EXPAND_WAVES, RDYN_EXPAND_WAVES))) void k_expand_staged(const RdynSweepArgs a)
{
  constexpr int MODE = MODE_REGRESSOR_EXPAND_STAGED;
  extern __shared__ __attribute__((aligned(16))) double expand_tile[];
  const unsigned blk = blockIdx.x;
#include "rdyn_local_sweep_body.inc"
}

// mixed-chain batch (BASELINE.json configs[4]): blockIdx.y selects one (chain, batch) item of a device table;
// the item descriptor and that chain's constants are wave-uniform, so both arrive by scalar loads -- the
// per-block "re-stage" of the chain costs a few s_load_dwordx16, no LDS and no barrier.
template <int NJ, int MODE>
__global__ __launch_bounds__(256) void k_local_sweep_multi(const RdynSweepArgs* __restrict__ table)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  const RDYN_CONST_AS RdynSweepArgs& a = *((const RDYN_CONST_AS RdynSweepArgs*)table + blockIdx.y);
#pragma clang diagnostic pop
  const unsigned blk = blockIdx.x;
  constexpr double* expand_tile = nullptr;
#include "rdyn_local_sweep_body.inc"
}

// ---------------------------------------------------------------------------------------------------
// Base-frame kinematics, stated as the reference states them.
// LEVEL: what the caller asked for -- 0 frames only (getTransformation(s)), 1 frames + the Jacobian (screws and origins kept), 2 frames +
// twists, 3 + spatial accelerations (round 6: the Jacobian belongs to level 1 alone -- carried through levels 2 and 3 its 12 NJ registers
// were dead weight in every getTwist / getDTwist call).  One instantiation per level: a getTransformation call does not pay for the velocity / acceleration
// recursions of getDTwist (750 -> 420 fp64 instructions per sample at 6 joints).
// STAGED (k_base_sweep_staged: 64-thread workgroups, `lds` = the wave's staging area): the sample-major records of a FULL wave leave
// through wave-private LDS in whole lines (rdyn_record_stage.h); a partial last wave keeps the 8-byte stores below.
template <int NJ, int LEVEL, bool STAGED, class Args>
__device__ __forceinline__ void base_sweep_body(const Args& a, const int64_t s, char* lds = nullptr)
{
  ChainPtr c = as_const(a.chain);
  const int lane = threadIdx.x & 63;
  bool stg = false;  // wave-uniform
  RecordRing<96> ringT;
  RecordRing<48> ringV, ringA;
  char* small_area = nullptr;
  if constexpr (STAGED)
  {
    const int64_t s_wave = s - lane;
    stg = a.n_samples - s_wave >= 64;
    if (stg)
    {
      char* p = lds;
      if (a.T_links)
      {
        ringT.init(p, a.T_links + s_wave * a.tl_ss, 96u * (NJ + 1), lane);
        p += RecordRing<96>::BYTES;
      }
      if (LEVEL >= 2 && a.twists)
      {
        ringV.init(p, a.twists + s_wave * a.tw_ss, 48u * (NJ + 1), lane);
        p += RecordRing<48>::BYTES;
      }
      if (LEVEL >= 3 && a.dtwists)
      {
        ringA.init(p, a.dtwists + s_wave * a.tw_ss, 48u * (NJ + 1), lane);
        p += RecordRing<48>::BYTES;
      }
      small_area = p;
    }
  }
  if (s >= a.n_samples) return;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = (LEVEL >= 2 && a.dq) ? a.dq + s * a.in_ss : nullptr;
  const double* __restrict__ ddqp = (LEVEL >= 3 && a.ddq) ? a.ddq + s * a.in_ss : nullptr;
  const int64_t es = a.out_se;  // element stride of every output record

  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3 p = mk(0, 0, 0);
  V3 vlin = mk(0, 0, 0), vang = mk(0, 0, 0), alin = mk(0, 0, 0), aang = mk(0, 0, 0);

  auto put3x4 = [&](double* __restrict__ o) {
    // column-major 3x4 [R | p]
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int r = 0; r < 3; ++r) o[(int64_t)(cc * 3 + r) * es] = R[r * 3 + cc];
    o[9 * es] = p.x;
    o[10 * es] = p.y;
    o[11 * es] = p.z;
  };
  auto put6 = [&](double* __restrict__ o, V3 l, V3 g) {
    o[0] = l.x; o[es] = l.y; o[2 * es] = l.z; o[3 * es] = g.x; o[4 * es] = g.y; o[5 * es] = g.z;
  };

  // the records that grow link by link: frame / twist / spatial acceleration of link `link` (0 = the base link)
  auto frame_out = [&](const int link) {
    if (!a.T_links) return;
    if (STAGED && stg)
    {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) ringT.put((uint32_t)(96 * link + 8 * (cc * 3 + r)), R[r * 3 + cc]);
      ringT.put((uint32_t)(96 * link + 72), p.x);
      ringT.put((uint32_t)(96 * link + 80), p.y);
      ringT.put((uint32_t)(96 * link + 88), p.z);
      ringT.flush(96 * link, 96 * (link + 1));
    }
    else
      put3x4(a.T_links + s * a.tl_ss + (int64_t)(12 * link) * es);
  };
  auto six_out = [&](const RecordRing<48>& ring, double* base, const int link, V3 l, V3 g) {
    if (STAGED && stg)
    {
      ring.put((uint32_t)(48 * link), l.x);
      ring.put((uint32_t)(48 * link + 8), l.y);
      ring.put((uint32_t)(48 * link + 16), l.z);
      ring.put((uint32_t)(48 * link + 24), g.x);
      ring.put((uint32_t)(48 * link + 32), g.y);
      ring.put((uint32_t)(48 * link + 40), g.z);
      ring.flush(48 * link, 48 * (link + 1));
    }
    else
      put6(base + s * a.tw_ss + (int64_t)(6 * link) * es, l, g);
  };
  // getJacobian, primitives_impl.h:939-945: column k = spatialTranslation(S_l, p_tool - p_l);
  // getJacobianLink, primitives_impl.h:951-979: the same referred to the origin of link j_link; only the FIRST
  // `up` input columns are filled, up = number of input joints upstream of the link (the reference's loop runs over
  // idx < joints.size() and reads m_active_joints.at(idx), :970-972) -- for the default, chain-ordered input list
  // these are exactly the link's parent joints.  j_link == NJ is the tool (up = n_active: plain getJacobian).
  // The reference point (origin of link j_link) is known only once the recursion has reached it: a FIRST PASS over the frames runs
  // up to that link (one sincos and ~70 fma per joint) and the sweep below forms every column as it passes its joint -- no axis and
  // origin kept per joint (12 NJ registers: 163 at 6 joints, round 5), the device of rdyn_long_kin.hip's Jacobian.
  V3 pref = mk(0, 0, 0);
  int j_up = 0;
  const bool jst = LEVEL == 1 && STAGED && stg && a.J;
  SmallRecords smJ;
  V3 zs[(LEVEL == 1 && STAGED) ? NJ : 1], pos[(LEVEL == 1 && STAGED) ? NJ : 1];
  if (LEVEL == 1 && !STAGED && a.J)
  {
    double R1[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3 p1 = mk(0, 0, 0);
#pragma unroll
    for (int l = 0; l < NJ; ++l)
    {
      JointRef J = c->j[l];
      if (l < a.j_link)
      {
        if (J.in_idx >= 0) ++j_up;
        const double ql = J.in_idx >= 0 ? qp[J.in_idx * a.in_sj] : 0.0;
        double Rpc[9];
        V3 t = ld3(J.t);
        if (J.type == RDYN_REVOLUTE)
        {
          double sn, cs;
          rdyn_sincos(ql, &sn, &cs);
          const double oc = 1.0 - cs;
#pragma unroll
          for (int i = 0; i < 9; ++i) Rpc[i] = fma(sn, J.B[i], fma(oc, J.C[i], J.A[i]));
        }
        else
        {
#pragma unroll
          for (int i = 0; i < 9; ++i) Rpc[i] = J.A[i];
          if (J.type == RDYN_PRISMATIC) t = axpy(t, ld3(J.up), ql);
        }
        p1 = p1 + rot(R1, t);
        double Rn[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            Rn[r * 3 + cc] = fma(R1[r * 3 + 0], Rpc[cc], fma(R1[r * 3 + 1], Rpc[3 + cc], R1[r * 3 + 2] * Rpc[6 + cc]));
#pragma unroll
        for (int i = 0; i < 9; ++i) R1[i] = Rn[i];
      }
    }
    pref = p1;
  }
  frame_out(0);
  if (LEVEL >= 2 && a.twists) six_out(ringV, a.twists, 0, vlin, vang);
  if (LEVEL >= 3 && a.dtwists) six_out(ringA, a.dtwists, 0, alin, aang);

#pragma unroll
  for (int f = 0; f < NJ; ++f)
  {
    JointRef J = c->j[f];
    const int type = J.type;
    const int idx = J.in_idx;
    double qf = 0.0, dqf = 0.0, ddqf = 0.0;
    if (idx >= 0)
    {
      const int64_t o = idx * a.in_sj;
      qf = qp[o];
      if (LEVEL >= 2 && dqp) dqf = dqp[o];
      if (LEVEL >= 3 && ddqp) ddqf = ddqp[o];
    }
    double Rpc[9];
    V3 t = ld3(J.t);
    if (type == RDYN_REVOLUTE)
    {
      double sn, cs;
      rdyn_sincos(qf, &sn, &cs);
      const double oc = 1.0 - cs;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rpc[i] = fma(sn, J.B[i], fma(oc, J.C[i], J.A[i]));
    }
    else
    {
#pragma unroll
      for (int i = 0; i < 9; ++i) Rpc[i] = J.A[i];
      if (type == RDYN_PRISMATIC) t = axpy(t, ld3(J.up), qf);
    }
    // screw axis of the child in the base frame, rotated by the PARENT frame (computeScrews, primitives_impl.h:879)
    const V3 zl = LEVEL >= 1 ? rot(R, ld3(J.up)) : mk(0, 0, 0);
    const V3 d = rot(R, t);  // p_l - p_{l-1}
    // T_bl[l] = T_bl[l-1] * T_pc   (computeFrames, primitives_impl.h:869)
    double Rn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Rn[r * 3 + cc] = fma(R[r * 3 + 0], Rpc[cc], fma(R[r * 3 + 1], Rpc[3 + cc], R[r * 3 + 2] * Rpc[6 + cc]));
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    p = p + d;
    if (LEVEL == 1 && STAGED)
    {
      zs[(LEVEL == 1 && STAGED) ? f : 0] = zl;
      pos[(LEVEL == 1 && STAGED) ? f : 0] = p;
    }
    if (LEVEL == 1 && !STAGED && a.J && idx >= 0)
    {
      // column idx of the Jacobian, as soon as its joint's axis and origin are known (the reference point came from the first pass)
      V3 jlin = mk(0, 0, 0), jang = mk(0, 0, 0);
      if (idx < j_up)
      {
        if (type == RDYN_REVOLUTE)
        {
          jlin = cross(zl, pref - p);
          jang = zl;
        }
        else if (type == RDYN_PRISMATIC)
          jlin = zl;
      }
      put6(a.J + s * a.j_ss + (int64_t)(6 * idx) * es, jlin, jang);
    }
    if (LEVEL >= 2)
    {
      // twists (getTwist, primitives_impl.h:1007-1008) and spatial accelerations (getDTwist, 1116-1117)
      V3 Sl = mk(0, 0, 0), Sa = mk(0, 0, 0);
      if (type == RDYN_REVOLUTE) Sa = zl;
      else if (type == RDYN_PRISMATIC) Sl = zl;
      const V3 nvl = axpy(vlin + cross(vang, d), Sl, dqf);
      const V3 nva = axpy(vang, Sa, dqf);
      if (LEVEL >= 3)
      {
        const V3 cl = cross(nva, Sl) + cross(nvl, Sa);  // spatialCrossProduct(v, S), sva.h:88-93
        const V3 ca = cross(nva, Sa);
        alin = axpy(axpy(alin + cross(aang, d), cl, dqf), Sl, ddqf);
        aang = axpy(axpy(aang, ca, dqf), Sa, ddqf);
      }
      vlin = nvl;
      vang = nva;
    }
    frame_out(f + 1);
    if (LEVEL >= 2 && a.twists) six_out(ringV, a.twists, f + 1, vlin, vang);
    if (LEVEL >= 3 && a.dtwists) six_out(ringA, a.dtwists, f + 1, alin, aang);
  }
  if (STAGED && stg)
  {
    if (a.T_links) ringT.finish();
    if (LEVEL >= 2 && a.twists) ringV.finish();
    if (LEVEL >= 3 && a.dtwists) ringA.finish();
  }
  SmallRecords sm;
  if (a.T_bt)
  {
    if (STAGED && stg)
    {
      sm.init(small_area, 12, lane);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) sm.put(cc * 3 + r, R[r * 3 + cc]);
      sm.put(9, p.x);
      sm.put(10, p.y);
      sm.put(11, p.z);
      sm.copy_out(a.T_bt + (s - lane) * a.tb_ss, lane);
    }
    else
      put3x4(a.T_bt + s * a.tb_ss);
  }
  if constexpr (LEVEL == 1 && STAGED)
  {
    if (a.J)
    {
      // the staged kernel (64-thread workgroups, its waves in flight bounded by the 19 KB record tile, not by registers) keeps every
      // joint's axis and origin and forms the columns here: the first pass of the two-pass form cost it 98 -> 134 us per 1e6
      if (jst) smJ.init(small_area, 6 * c->n_active, lane);
      V3 pr = p;
      int up = 0;
#pragma unroll
      for (int l = 0; l < NJ; ++l)
      {
        if (a.j_link == l + 1) pr = pos[l];
        if (l < a.j_link && c->j[l].in_idx >= 0) ++up;
      }
      if (a.j_link == 0) pr = mk(0, 0, 0);
#pragma unroll
      for (int l = 0; l < NJ; ++l)
      {
        const int k = c->j[l].in_idx;
        if (k < 0) continue;
        const int ty = c->j[l].type;
        V3 jlin = mk(0, 0, 0), jang = mk(0, 0, 0);
        if (k < up)
        {
          if (ty == RDYN_REVOLUTE)
          {
            jlin = cross(zs[l], pr - pos[l]);
            jang = zs[l];
          }
          else if (ty == RDYN_PRISMATIC)
            jlin = zs[l];
        }
        if (jst)
        {
          smJ.put(6 * k, jlin.x); smJ.put(6 * k + 1, jlin.y); smJ.put(6 * k + 2, jlin.z);
          smJ.put(6 * k + 3, jang.x); smJ.put(6 * k + 4, jang.y); smJ.put(6 * k + 5, jang.z);
        }
        else
          put6(a.J + s * a.j_ss + (int64_t)(6 * k) * es, jlin, jang);
      }
      if (jst) smJ.copy_out(a.J + (s - lane) * a.j_ss, lane);
    }
  }
}

template <int NJ, int LEVEL>
__global__ __launch_bounds__(256) void k_base_sweep(const RdynKinArgs a)
{
  base_sweep_body<NJ, LEVEL, false>(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}
// the sample-major records in whole lines: one wave per workgroup, its staging area in dynamic LDS (base_stage_lds_bytes)
template <int NJ, int LEVEL>
__global__ __launch_bounds__(64) void k_base_sweep_staged(const RdynKinArgs a)
{
  extern __shared__ __attribute__((aligned(16))) char stage_lds[];
  base_sweep_body<NJ, LEVEL, true>(a, (int64_t)blockIdx.x * 64 + threadIdx.x, stage_lds);
}

// Every getter of a sample in ONE launch (rdyn_evaluate_all): blockIdx.y picks a role -- frames of all links, the tool Jacobian, twists +
// spatial accelerations (the three levels of the base-frame sweep), joint torque, its non-linear part, joint inertia, regressor (the
// four modes of the local-frame sweep) -- the same device code as the single-purpose kernels, side by side on the chip.  Made for
// SMALL batches: one sample per call through the C++ facade costs a launch and a round trip, not seven.
template <int NJ>
__global__ __launch_bounds__(256) void k_sample_all(const RdynAllArgs all)
{
  const unsigned blk = blockIdx.x;
  constexpr double* expand_tile = nullptr;
  switch (blockIdx.y)
  {
  case 0:
    if (all.frames.T_bt || all.frames.T_links) base_sweep_body<NJ, 0, false>(all.frames, (int64_t)blk * 256 + threadIdx.x);
    return;
  case 1:
    if (all.jacobian.J) base_sweep_body<NJ, 1, false>(all.jacobian, (int64_t)blk * 256 + threadIdx.x);
    return;
  case 2:
    if (all.twists.dtwists) base_sweep_body<NJ, 3, false>(all.twists, (int64_t)blk * 256 + threadIdx.x);
    else if (all.twists.twists) base_sweep_body<NJ, 2, false>(all.twists, (int64_t)blk * 256 + threadIdx.x);
    return;
  case 3:
  {
    if (!all.torque.tau) return;
    constexpr int MODE = MODE_TORQUE;
    const RdynSweepArgs& a = all.torque;
#include "rdyn_local_sweep_body.inc"
    return;
  }
  case 4:
  {
    if (!all.torque_nl.tau) return;
    constexpr int MODE = MODE_TORQUE;
    const RdynSweepArgs& a = all.torque_nl;
#include "rdyn_local_sweep_body.inc"
    return;
  }
  case 5:
  {
    if (!all.inertia.M) return;
    constexpr int MODE = MODE_INERTIA;
    const RdynSweepArgs& a = all.inertia;
#include "rdyn_local_sweep_body.inc"
    return;
  }
  default:
  {
    if (!all.regressor.Y) return;
    constexpr int MODE = MODE_REGRESSOR;
    const RdynSweepArgs& a = all.regressor;
#include "rdyn_local_sweep_body.inc"
    return;
  }
  }
}

template <int NJ>
hipError_t launch_local_nj(int mode, const RdynSweepArgs& a, hipStream_t st)
{
  const unsigned grid = (unsigned)((a.n_samples + 255) / 256);
  switch (mode)
  {
  case MODE_REGRESSOR_EXPAND_STAGED:
    hipLaunchKernelGGL((k_expand_staged<NJ>), dim3((unsigned)((a.n_samples + 63) / 64)), dim3(64), (size_t)64 * (RDYN_EXPAND_TILE_COLS(NJ) * NJ + 2 + 3 * NJ + 1) * sizeof(double), st, a);
    break;
  case MODE_REGRESSOR: hipLaunchKernelGGL((k_local_sweep<NJ, MODE_REGRESSOR>), dim3(grid), dim3(256), 0, st, a); break;
  case MODE_REGRESSOR_GRAM: hipLaunchKernelGGL((k_local_sweep<NJ, MODE_REGRESSOR_GRAM>), dim3(grid), dim3(256), 0, st, a); break;
  case MODE_REGRESSOR_EXPAND: hipLaunchKernelGGL((k_local_sweep<NJ, MODE_REGRESSOR_EXPAND>), dim3(grid), dim3(256), 0, st, a); break;
  // (a.staged = doubles per record: the sample-major records through one LDS tile per wave, 64 (rec | 1) doubles)
  case MODE_TORQUE:
    if (a.staged) hipLaunchKernelGGL((k_local_sweep_rec<NJ, MODE_TORQUE>), dim3((unsigned)((a.n_samples + 63) / 64)), dim3(64), (size_t)64 * (a.staged | 1) * 8, st, a);
    else hipLaunchKernelGGL((k_local_sweep<NJ, MODE_TORQUE>), dim3(grid), dim3(256), 0, st, a);
    break;
  default:
    if (a.staged) hipLaunchKernelGGL((k_local_sweep_rec<NJ, MODE_INERTIA>), dim3((unsigned)((a.n_samples + 63) / 64)), dim3(64), (size_t)32 * (a.staged | 1) * 8, st, a);  // a half-wave tile
    else hipLaunchKernelGGL((k_local_sweep<NJ, MODE_INERTIA>), dim3(grid), dim3(256), 0, st, a);
    break;
  }
  return hipGetLastError();
}
constexpr size_t kFramesStageLds = 33 * 1024;  // 160 KB / 33 KB: four waves per CU
template <int NJ>
hipError_t launch_base_nj(const RdynKinArgs& a, hipStream_t st)
{
  if (a.staged)
  {
    // rings in the order the kernel lays them out, then the tile of the records that are complete at the end of the sweep
    size_t lds = 0, small = 0;
    if (a.T_links) lds += RecordRing<96>::BYTES;
    if (a.twists) lds += RecordRing<48>::BYTES;
    if (a.dtwists) lds += RecordRing<48>::BYTES;
    if (a.T_bt) small = (size_t)64 * 13 * 8;
    if (a.J && (size_t)64 * (size_t)((6 * a.n_active) | 1) * 8 > small) small = (size_t)64 * (size_t)((6 * a.n_active) | 1) * 8;
    lds += small;
    // all frames (720 B per sample at 6 joints): FOUR waves per CU are the optimum -- 118-122 us per 1e6 against 129-133 at five to seven
    // (tools/build_variant.sh ... -DRDYN_STAGE_LDS_PAD, profiles/r6/stage_occupancy.txt; the twist levels, a third of the bytes per wave
    // behind a longer recursion, want every wave they can get: 90 / 99 / 122 us at 8 / 6 / 4 waves) -- asked for through the LDS request
    if (a.T_links && lds < kFramesStageLds) lds = kFramesStageLds;
#ifdef RDYN_STAGE_LDS_PAD  // timing experiment: fewer waves per CU through the LDS request
    lds += RDYN_STAGE_LDS_PAD;
#endif
    const unsigned g64 = (unsigned)((a.n_samples + 63) / 64);
    if (a.dtwists) hipLaunchKernelGGL((k_base_sweep_staged<NJ, 3>), dim3(g64), dim3(64), lds, st, a);
    else if (a.twists) hipLaunchKernelGGL((k_base_sweep_staged<NJ, 2>), dim3(g64), dim3(64), lds, st, a);
    else if (a.J) hipLaunchKernelGGL((k_base_sweep_staged<NJ, 1>), dim3(g64), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((k_base_sweep_staged<NJ, 0>), dim3(g64), dim3(64), lds, st, a);
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((a.n_samples + 255) / 256);
  // the cheapest instantiation that produces everything asked for
  if (a.dtwists) hipLaunchKernelGGL((k_base_sweep<NJ, 3>), dim3(grid), dim3(256), 0, st, a);
  else if (a.twists) hipLaunchKernelGGL((k_base_sweep<NJ, 2>), dim3(grid), dim3(256), 0, st, a);
  else if (a.J) hipLaunchKernelGGL((k_base_sweep<NJ, 1>), dim3(grid), dim3(256), 0, st, a);
#ifdef RDYN_BASE0_LDS_PAD  // timing experiment: fewer waves per CU for the frames-only level through an LDS request
  else hipLaunchKernelGGL((k_base_sweep<NJ, 0>), dim3(grid), dim3(256), RDYN_BASE0_LDS_PAD, st, a);
#else
  else hipLaunchKernelGGL((k_base_sweep<NJ, 0>), dim3(grid), dim3(256), 0, st, a);
#endif
  return hipGetLastError();
}

}  // namespace

#define RDYN_DISPATCH_NJ(nj, CALL)                 \
  switch (nj)                                      \
  {                                                \
  case 1: return CALL(1);                          \
  case 2: return CALL(2);                          \
  case 3: return CALL(3);                          \
  case 4: return CALL(4);                          \
  case 5: return CALL(5);                          \
  case 6: return CALL(6);                          \
  case 7: return CALL(7);                          \
  case 8: return CALL(8);                          \
  case 9: return CALL(9);                          \
  case 10: return CALL(10);                        \
  default: return hipErrorInvalidValue;            \
  }

// The file is compiled in slices (Makefile: -DRDYN_KERNELS_PART=0..3, like rdyn_image_part.hip) so that the instantiations build in
// parallel: 0 the single-chain sweeps, 1 every getter of a sample in one launch, 2 the base sweeps, 3 the mixed-chain plans.
#ifndef RDYN_KERNELS_PART
#error "compile with -DRDYN_KERNELS_PART=<0..3>"
#endif
#if RDYN_KERNELS_PART == 0
hipError_t rdyn_launch_local_sweep(int n_joints, int mode, const RdynSweepArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
#define CALL(N) launch_local_nj<N>(mode, a, st)
  RDYN_DISPATCH_NJ(n_joints, CALL)
#undef CALL
}
#endif

#if RDYN_KERNELS_PART == 1
namespace
{
template <int NJ>
hipError_t launch_all_nj(const RdynAllArgs& a, hipStream_t st)
{
  hipLaunchKernelGGL((k_sample_all<NJ>), dim3((unsigned)((a.n_samples + 255) / 256), 7), dim3(256), 0, st, a);
  return hipGetLastError();
}
}  // namespace
hipError_t rdyn_launch_sample_all(int n_joints, const RdynAllArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
#define CALL(N) launch_all_nj<N>(a, st)
  RDYN_DISPATCH_NJ(n_joints, CALL)
#undef CALL
}
#endif

#if RDYN_KERNELS_PART == 2
static hipError_t launch_base_dispatch(int n_joints, const RdynKinArgs& a, hipStream_t st)
{
#define CALL(N) launch_base_nj<N>(a, st)
  RDYN_DISPATCH_NJ(n_joints, CALL)
#undef CALL
}
hipError_t rdyn_launch_base_sweep(int n_joints, const RdynKinArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  if (a.J && (a.twists || a.dtwists || a.T_bt || a.T_links))
  {
    // the Jacobian is level 1's alone (the twist levels do not keep the screws and origins it needs): two launches
    RdynKinArgs j = a, rest = a;
    j.twists = j.dtwists = j.T_bt = j.T_links = nullptr;
    rest.J = nullptr;
    const hipError_t e = launch_base_dispatch(n_joints, j, st);
    if (e != hipSuccess) return e;
    return launch_base_dispatch(n_joints, rest, st);
  }
  return launch_base_dispatch(n_joints, a, st);
}
#endif

#if RDYN_KERNELS_PART == 3
namespace
{
template <int NJ>
hipError_t launch_local_multi_nj(int mode, const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st)
{
  const dim3 grid((unsigned)((max_samples + 255) / 256), (unsigned)n_items);
  switch (mode)
  {
  case MODE_REGRESSOR: hipLaunchKernelGGL((k_local_sweep_multi<NJ, MODE_REGRESSOR>), grid, dim3(256), 0, st, table); break;
  case MODE_TORQUE: hipLaunchKernelGGL((k_local_sweep_multi<NJ, MODE_TORQUE>), grid, dim3(256), 0, st, table); break;
  default: hipLaunchKernelGGL((k_local_sweep_multi<NJ, MODE_INERTIA>), grid, dim3(256), 0, st, table); break;
  }
  return hipGetLastError();
}
}  // namespace

// `table` = device array of n_items descriptors, all for chains with `n_joints` joints; max_samples = largest batch
hipError_t rdyn_launch_local_sweep_multi(int n_joints, int mode, const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st)
{
  if (n_items <= 0 || max_samples <= 0) return hipSuccess;
#define CALL(N) launch_local_multi_nj<N>(mode, table, n_items, max_samples, st)
  RDYN_DISPATCH_NJ(n_joints, CALL)
#undef CALL
}
#endif
