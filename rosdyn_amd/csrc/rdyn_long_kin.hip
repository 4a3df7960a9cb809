// rdyn_long_kin.hip -- base-frame kinematics of chains LONGER than the unrolled kernels sweep (11 .. RDYN_MAX_JOINTS chain joints).
//
// The reference's default build has no bound on the chain length (rosdyn_core/CMakeLists.txt:12-16, internal/types.h:120-129); the
// kernels of rdyn_kernels.hip / rdyn_kin_ext.hip are instantiated per joint count with every per-link quantity in registers.  The
// by-link kinematic outputs need none of that: frames, twists and their derivatives are RUNNING quantities of a base -> tool
// recursion, one record per link streamed out as it is finished.  Here the link loop is rolled (run-time trip count, joint f's
// constants by scalar loads at a wave-uniform offset into RdynLongChainConst), any number of input joints in any order:
//   k_long_base<LEVEL>   getTransformation(s) :863-912, getJacobian / getJacobianLink :927-979, getTwist :981-1013, getDTwist :1082-1124
//   k_long_ext<WRENCH>   getDTwistLinearPart / NonLinearPart :1029-1080, getDDTwist* :1126-1223, getWrench :1225-1262,
//                        getJointTorque with external wrenches :1264-1272
// (paths under /root/reference/rosdyn_core/include/rosdyn_core/internal/primitives_impl.h).  Same arithmetic, statement by
// statement, as the unrolled kernels: the parity tests hold both against the same oracle.
//
// Jacobian: column k needs the reference point (origin of the requested link), known only once the recursion has reached it; the
// frames are cheap (one sincos and ~70 fma per joint), so a first pass runs them up to that link and the main pass forms the columns
// as it goes -- no per-joint storage.  Wrench: two forward passes (the total, then total - upstream), see k_long_ext.
#include <hip/hip_runtime.h>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_record_stage.h"

namespace
{
typedef const RDYN_CONST_AS RdynLongChainConst* LongChainPtr;
__device__ __forceinline__ LongChainPtr as_const_long(const RdynLongChainConst* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (LongChainPtr)p;
#pragma clang diagnostic pop
}

struct S6
{
  V3 l, a;
};
__device__ __forceinline__ S6 xs(S6 v, S6 s)  // spatialCrossProduct, spacevect_algebra.h:88-93
{
  S6 r;
  r.a = cross(v.a, s.a);
  r.l = cross(v.a, s.l) + cross(v.l, s.a);
  return r;
}
__device__ __forceinline__ S6 shift(S6 t, V3 d)  // spatialTranslation, spacevect_algebra.h:129-133
{
  S6 r;
  r.l = t.l + cross(t.a, d);
  r.a = t.a;
  return r;
}
__device__ __forceinline__ S6 axpy6(S6 a, S6 b, double s)
{
  S6 r;
  r.l = axpy(a.l, b.l, s);
  r.a = axpy(a.a, b.a, s);
  return r;
}

// One step of computeFrames / computeScrews (primitives_impl.h:863-882): on entry R, p = frame of the parent link; on exit of the
// child.  zl = the joint axis in the base frame (rotated by the PARENT frame, :879), d = p_child - p_parent.
__device__ __forceinline__ void frame_step(JointRef J, double qf, double (&R)[9], V3& p, V3& zl, V3& d)
{
  const int type = J.type;
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
  zl = rot(R, ld3(J.up));
  d = rot(R, t);
  double Rn[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) Rn[r * 3 + cc] = fma(R[r * 3 + 0], Rpc[cc], fma(R[r * 3 + 1], Rpc[3 + cc], R[r * 3 + 2] * Rpc[6 + cc]));
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Rn[i];
  p = p + d;
}

// LEVEL as in k_base_sweep: 0 frames only, 1 + the Jacobian, 2 + twists, 3 + spatial accelerations
// STAGED (a.staged: sample-major records at their natural stride, line-aligned outputs): 64-thread workgroups; the records of a FULL wave
// leave through wave-private LDS in whole lines (rdyn_record_stage.h: one ring per by-link output, a tile for the tool frame / Jacobian)
template <int LEVEL, bool STAGED>
__global__ __launch_bounds__(STAGED ? 64 : 256) void k_long_base(const RdynKinArgs a)
{
  constexpr int BS = STAGED ? 64 : 256;
  extern __shared__ __attribute__((aligned(16))) char stage_lds[];
  LongChainPtr c = as_const_long(a.chain_long);
  const int nj = c->n_joints;
  const int64_t s = (int64_t)blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool stg = STAGED && a.n_samples - (int64_t)blockIdx.x * BS >= 64;  // wave-uniform
  RecordRing<96> ringT;
  RecordRing<48> ringV, ringA;
  SmallRecords smJ;
  char* small_area = nullptr;
  if constexpr (STAGED)
  {
    if (stg)
    {
      const int64_t s_wave = (int64_t)blockIdx.x * BS;
      char* lp = stage_lds;
      if (a.T_links)
      {
        ringT.init(lp, a.T_links + s_wave * a.tl_ss, 96u * (uint32_t)(nj + 1), lane);
        lp += RecordRing<96>::BYTES;
      }
      if (LEVEL >= 2 && a.twists)
      {
        ringV.init(lp, a.twists + s_wave * a.tw_ss, 48u * (uint32_t)(nj + 1), lane);
        lp += RecordRing<48>::BYTES;
      }
      if (LEVEL >= 3 && a.dtwists)
      {
        ringA.init(lp, a.dtwists + s_wave * a.tw_ss, 48u * (uint32_t)(nj + 1), lane);
        lp += RecordRing<48>::BYTES;
      }
      small_area = lp;
      if (LEVEL >= 1 && a.J) smJ.init(small_area, 6 * a.n_active, lane);
    }
  }
  if (s >= a.n_samples) return;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = (LEVEL >= 2 && a.dq) ? a.dq + s * a.in_ss : nullptr;
  const double* __restrict__ ddqp = (LEVEL >= 3 && a.ddq) ? a.ddq + s * a.in_ss : nullptr;
  const int64_t es = a.out_se;

  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3 p = mk(0, 0, 0);
  auto put3x4 = [&](double* __restrict__ o) {
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

  // Jacobian: origin of the reference link (j_link == nj: the tool) by a first pass over the frames upstream of it
  V3 pref = mk(0, 0, 0);
  if (LEVEL >= 1 && a.J)
  {
#pragma unroll 1
    for (int f = 0; f < a.j_link; ++f)
    {
      JointRef J = c->j[f];
      const int idx = J.in_idx;
      V3 zl, d;
      frame_step(J, idx >= 0 ? qp[idx * a.in_sj] : 0.0, R, p, zl, d);
    }
    pref = p;
    p = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }

  V3 vlin = mk(0, 0, 0), vang = mk(0, 0, 0), alin = mk(0, 0, 0), aang = mk(0, 0, 0);
  auto frame_out = [&](const int link) {
    if (!a.T_links) return;
    if (STAGED && stg)
    {
      const uint32_t x0 = 96u * (uint32_t)link;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) ringT.put(x0 + 8u * (cc * 3 + r), R[r * 3 + cc]);
      ringT.put(x0 + 72u, p.x);
      ringT.put(x0 + 80u, p.y);
      ringT.put(x0 + 88u, p.z);
      ringT.flush(96 * link, 96 * (link + 1));
    }
    else
      put3x4(a.T_links + s * a.tl_ss + (int64_t)(12 * link) * es);
  };
  auto six_out = [&](const RecordRing<48>& ring, double* base, const int link, V3 l, V3 g) {
    if (STAGED && stg)
    {
      const uint32_t x0 = 48u * (uint32_t)link;
      ring.put(x0, l.x);
      ring.put(x0 + 8u, l.y);
      ring.put(x0 + 16u, l.z);
      ring.put(x0 + 24u, g.x);
      ring.put(x0 + 32u, g.y);
      ring.put(x0 + 40u, g.z);
      ring.flush(48 * link, 48 * (link + 1));
    }
    else
      put6(base + s * a.tw_ss + (int64_t)(6 * link) * es, l, g);
  };
  frame_out(0);
  if (LEVEL >= 2 && a.twists) six_out(ringV, a.twists, 0, vlin, vang);
  if (LEVEL >= 3 && a.dtwists) six_out(ringA, a.dtwists, 0, alin, aang);
#pragma unroll 1
  for (int f = 0; f < nj; ++f)
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
    V3 zl, d;
    frame_step(J, qf, R, p, zl, d);
    if (LEVEL >= 2)
    {
      // twists (getTwist, :1007-1008) and spatial accelerations (getDTwist, :1116-1117)
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
    if (LEVEL >= 1 && a.J && idx >= 0)
    {
      // getJacobian :939-945 / getJacobianLink :951-979: column k = spatialTranslation(S_l, p_ref - p_l) for the FIRST j_up input
      // columns (j_up = input joints upstream of the link, counted by the host), zero beyond
      V3 jlin = mk(0, 0, 0), jang = mk(0, 0, 0);
      if (idx < a.j_up)
      {
        if (type == RDYN_REVOLUTE)
        {
          jlin = cross(zl, pref - p);
          jang = zl;
        }
        else if (type == RDYN_PRISMATIC)
          jlin = zl;
      }
      if (STAGED && stg)
      {
        smJ.put(6 * idx, jlin.x); smJ.put(6 * idx + 1, jlin.y); smJ.put(6 * idx + 2, jlin.z);
        smJ.put(6 * idx + 3, jang.x); smJ.put(6 * idx + 4, jang.y); smJ.put(6 * idx + 5, jang.z);
      }
      else
        put6(a.J + s * a.j_ss + (int64_t)(6 * idx) * es, jlin, jang);
    }
  }
  if (STAGED && stg)
  {
    if (a.T_links) ringT.finish();
    if (LEVEL >= 2 && a.twists) ringV.finish();
    if (LEVEL >= 3 && a.dtwists) ringA.finish();
    if (LEVEL >= 1 && a.J) smJ.copy_out(a.J + (s - lane) * a.j_ss, lane);
  }
  if (a.T_bt)
  {
    if (STAGED && stg)
    {
      SmallRecords sm;
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
}

// The split / jerk recursions (WRENCH = false) and the wrench recursion (true).  The reference sums the link wrenches tool -> base
// (primitives_impl.h:1231-1259); a forward sweep knows a link's own wrench only when it gets there, and parking one record per link
// and lane in LDS (as the unrolled kernel does) leaves two waves per CU at 14 links.  Here the sweep runs TWICE: the first pass adds
// up every link's own wrench about the base origin, the second forms w[l] = total - (the links upstream of l) as it goes, refers it to
// the link's origin, and reads the joint torque off it -- no per-link storage, full occupancy; the arithmetic differs from the suffix
// sums by the rounding of one subtraction (1e-16 of the chain's total wrench).
// STAGED: as k_long_base -- one ring per requested record (the link wrenches leave in link order in the second pass), the joint torques
// through a tile.
template <bool WRENCH, bool STAGED>
__global__ __launch_bounds__(STAGED ? 64 : 256) void k_long_ext(const RdynKinExtArgs a)
{
  constexpr int BS = STAGED ? 64 : 256;
  extern __shared__ __attribute__((aligned(16))) char stage_lds[];
  LongChainPtr c = as_const_long(a.chain_long);
  const int nj = c->n_joints;
  const int64_t s = (int64_t)blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool stg = STAGED && a.n_samples - (int64_t)blockIdx.x * BS >= 64;  // wave-uniform
  // rings: WRENCH -- [0] the link wrenches; otherwise one per requested split / jerk output
  RecordRing<48> rings[STAGED ? (WRENCH ? 1 : 5) : 1];
  SmallRecords smT;
  bool tau_staged = false;
  if constexpr (STAGED)
  {
    if (stg)
    {
      const int64_t s_wave = (int64_t)blockIdx.x * BS;
      char* lp = stage_lds;
      if constexpr (WRENCH)
      {
        if (a.wrench)
        {
          rings[0].init(lp, a.wrench + s_wave * a.out_ss, 48u * (uint32_t)(nj + 1), lane);
          lp += RecordRing<48>::BYTES;
        }
        if (a.tau && a.tau_sj == 1 && a.tau_ss == c->n_active && (((uintptr_t)a.tau) & 127u) == 0)
        {
          smT.init(lp, c->n_active, lane);
          tau_staged = true;
        }
      }
      else
      {
        double* const outs[5] = {a.dtw_lin, a.dtw_nonlin, a.ddtw, a.ddtw_lin, a.ddtw_nonlin};
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (outs[k])
          {
            rings[(STAGED && !WRENCH) ? k : 0].init(lp, outs[k] + s_wave * a.out_ss, 48u * (uint32_t)(nj + 1), lane);
            lp += RecordRing<48>::BYTES;
          }
      }
    }
  }
  if (s >= a.n_samples) return;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = a.dq ? a.dq + s * a.in_ss : nullptr;
  const double* __restrict__ ddqp = a.ddq ? a.ddq + s * a.in_ss : nullptr;
  const double* __restrict__ dddqp = a.dddq ? a.dddq + s * a.in_ss : nullptr;
  const int64_t es = a.out_se;
  auto put6 = [&](double* base, int64_t first_elem, S6 x) {
    double* __restrict__ o = base + s * a.out_ss + first_elem * es;
    o[0] = x.l.x; o[es] = x.l.y; o[2 * es] = x.l.z; o[3 * es] = x.a.x; o[4 * es] = x.a.y; o[5 * es] = x.a.z;
  };
  const S6 zero = {mk(0, 0, 0), mk(0, 0, 0)};
  const V3 grav = mk(c->g[0], c->g[1], c->g[2]);
  auto ext_of = [&](int link) -> S6 {  // -ext_wrenches_in_link_frame.at(link), :1255
    S6 e = zero;
    if (a.ext)
    {
      const double* __restrict__ ep = a.ext + s * a.ext_ss + (int64_t)(6 * link) * a.ext_se;
      e.l = mk(-ep[0], -ep[a.ext_se], -ep[2 * a.ext_se]);
      e.a = mk(-ep[3 * a.ext_se], -ep[4 * a.ext_se], -ep[5 * a.ext_se]);
    }
    return e;
  };
  // link `link` of output k (WRENCH: k = 0, the link wrenches)
  auto out6 = [&](const int k, double* base, const int link, S6 x) {
    if (!base) return;
    if (STAGED && stg)
    {
      const RecordRing<48>& ring = rings[(STAGED && !WRENCH) ? k : 0];
      const uint32_t x0 = 48u * (uint32_t)link;
      ring.put(x0, x.l.x);
      ring.put(x0 + 8u, x.l.y);
      ring.put(x0 + 16u, x.l.z);
      ring.put(x0 + 24u, x.a.x);
      ring.put(x0 + 32u, x.a.y);
      ring.put(x0 + 40u, x.a.z);
      ring.flush(48 * link, 48 * (link + 1));
    }
    else
      put6(base, 6 * link, x);
  };
  if (!WRENCH)
  {
    out6(0, a.dtw_lin, 0, zero);
    out6(1, a.dtw_nonlin, 0, zero);
    out6(2, a.ddtw, 0, zero);
    out6(3, a.ddtw_lin, 0, zero);
    out6(4, a.ddtw_nonlin, 0, zero);
  }
  S6 total = zero;
  double* __restrict__ tp = (WRENCH && a.tau) ? a.tau + s * a.tau_ss : nullptr;
#pragma unroll 1
  for (int pass = 0; pass < (WRENCH ? 2 : 1); ++pass)
  {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3 p = mk(0, 0, 0);
    S6 v = zero, acc = zero, aL = zero, aN = zero, jk = zero, jL = zero, jN = zero;
    S6 upstream = zero;  // second pass: own wrenches of the links before the current one, about the base origin
    if (WRENCH)
    {
      const S6 own0 = ext_of(0);  // spatialTranformation(-ext, identity); no inertial / gravity term on the base link (:1233-1237)
      if (pass == 0)
        total = own0;
      else
      {
        out6(0, a.wrench, 0, total);  // the base link carries the whole chain; its origin IS the base origin
        upstream = own0;
      }
    }
#pragma unroll 1
    for (int f = 0; f < nj; ++f)
    {
      JointRef J = c->j[f];
      const int type = J.type;
      const int idx = J.in_idx;
      double qf = 0.0, dqf = 0.0, ddqf = 0.0, dddqf = 0.0;
      if (idx >= 0)
      {
        const int64_t o = idx * a.in_sj;
        qf = qp[o];
        if (dqp) dqf = dqp[o];
        if (ddqp) ddqf = ddqp[o];
        if (dddqp) dddqf = dddqp[o];
      }
      V3 zl, d;
      frame_step(J, qf, R, p, zl, d);
      S6 S = zero;
      if (type == RDYN_REVOLUTE) S.a = zl;
      else if (type == RDYN_PRISMATIC) S.l = zl;
      v = axpy6(shift(v, d), S, dqf);
      const S6 vxs = xs(v, S);
      acc = axpy6(axpy6(shift(acc, d), vxs, dqf), S, ddqf);  // getDTwist, :1116-1117
      if (!WRENCH)
      {
        aL = axpy6(shift(aL, d), S, ddqf);
        aN = axpy6(shift(aN, d), vxs, dqf);
        const S6 axs = xs(acc, S), vvxs = xs(v, vxs);
        S6 cq;
        cq.l = axs.l + vvxs.l;
        cq.a = axs.a + vvxs.a;
        jk = axpy6(axpy6(axpy6(shift(jk, d), S, dddqf), vxs, ddqf), cq, dqf);
        jL = axpy6(shift(jL, d), S, dddqf);
        jN = axpy6(axpy6(shift(jN, d), vxs, ddqf), cq, dqf);
        out6(0, a.dtw_lin, f + 1, aL);
        out6(1, a.dtw_nonlin, f + 1, aN);
        out6(2, a.ddtw, f + 1, jk);
        out6(3, a.ddtw_lin, f + 1, jL);
        out6(4, a.ddtw_nonlin, f + 1, jN);
      }
      else
      {
        // link f + 1: spatial inertia about its origin from the nominal parameters [m, m c, Io] (primitives_impl.h:399-417)
        const double m = J.pi[0];
        const V3 mc = mk(J.pi[1], J.pi[2], J.pi[3]);
        auto Imul = [&](S6 x) -> S6 {  // [[m 1, m c^T],[m c^, Io]] x   (spacevect_algebra.h:232-239)
          S6 r;
          r.l = mk(m * x.l.x, m * x.l.y, m * x.l.z) - cross(mc, x.a);
          r.a = cross(mc, x.l) + symv(J.pi + 4, x.a);
          return r;
        };
        S6 al, vloc;
        al.l = rotT(R, acc.l); al.a = rotT(R, acc.a);      // spatialRotation(m_Dtwists, R^T), :1242
        vloc.l = rotT(R, v.l); vloc.a = rotT(R, v.a);      // :1245
        const S6 Iv = Imul(vloc), Ia = Imul(al);
        S6 wl;  // I a + v x* (I v), spatialDualCrossProduct spacevect_algebra.h:108-113
        wl.l = Ia.l + cross(vloc.a, Iv.l);
        wl.a = Ia.a + cross(vloc.a, Iv.a) + cross(vloc.l, Iv.l);
        S6 own;
        own.l = rot(R, wl.l);                              // :1248
        own.a = rot(R, wl.a);
        own.l = own.l - mk(m * grav.x, m * grav.y, m * grav.z);   // gravity wrench, :1249-1250
        own.a = own.a - cross(rot(R, mc), grav);
        const S6 e = ext_of(f + 1);                        // spatialTranformation(-ext, T_bl): twist form, :1255 / spacevect_algebra.h:193-197
        const V3 Ra = rot(R, e.a);
        own.l = own.l + rot(R, e.l) + cross(Ra, p);
        own.a = own.a + Ra;
        own.a = own.a + cross(p, own.l);  // referred to the base origin: sums need no per-pair translation
        if (pass == 0)
        {
          total.l = total.l + own.l;
          total.a = total.a + own.a;
        }
        else
        {
          // w[f + 1] = the links f + 1 .. tool, referred to link f + 1's origin (spatialDualTranslation: ang += lin x d, :1255);
          // tau of joint f = w[f + 1] . the joint's screw at that origin (:1264-1272)
          S6 w;
          w.l = total.l - upstream.l;
          w.a = (total.a - upstream.a) - cross(p, w.l);
          out6(0, a.wrench, f + 1, w);
          if (tp && idx >= 0)
          {
            const double tq = type == RDYN_REVOLUTE ? dot(zl, w.a) : (type == RDYN_PRISMATIC ? dot(zl, w.l) : 0.0);
            if (STAGED && tau_staged) smT.put(idx, tq);
            else tp[idx * a.tau_sj] = tq;
          }
          upstream.l = upstream.l + own.l;
          upstream.a = upstream.a + own.a;
        }
      }
    }
  }
  if (STAGED && stg)
  {
    if constexpr (WRENCH)
    {
      if (a.wrench) rings[0].finish();
      if (tau_staged) smT.copy_out(a.tau + (s - lane) * a.tau_ss, lane);
    }
    else
    {
      double* const outs[5] = {a.dtw_lin, a.dtw_nonlin, a.ddtw, a.ddtw_lin, a.ddtw_nonlin};
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (outs[k]) rings[(STAGED && !WRENCH) ? k : 0].finish();
    }
  }
}

}  // namespace

hipError_t rdyn_launch_long_base(const RdynKinArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  if (a.staged)
  {
    size_t lds = 0, small = 0;
    if (a.T_links) lds += RecordRing<96>::BYTES;
    if (a.twists) lds += RecordRing<48>::BYTES;
    if (a.dtwists) lds += RecordRing<48>::BYTES;
    if (a.T_bt) small = (size_t)64 * 13 * 8;
    if (a.J && (size_t)64 * (size_t)((6 * a.n_active) | 1) * 8 > small) small = (size_t)64 * (size_t)((6 * a.n_active) | 1) * 8;
    lds += small;
    if (lds <= 64 * 1024)
    {
      const unsigned g64 = (unsigned)((a.n_samples + 63) / 64);
      if (a.dtwists) hipLaunchKernelGGL((k_long_base<3, true>), dim3(g64), dim3(64), lds, st, a);
      else if (a.twists) hipLaunchKernelGGL((k_long_base<2, true>), dim3(g64), dim3(64), lds, st, a);
      else if (a.J) hipLaunchKernelGGL((k_long_base<1, true>), dim3(g64), dim3(64), lds, st, a);
      else hipLaunchKernelGGL((k_long_base<0, true>), dim3(g64), dim3(64), lds, st, a);
      return hipGetLastError();
    }
  }
  const unsigned grid = (unsigned)((a.n_samples + 255) / 256);
  if (a.dtwists) hipLaunchKernelGGL((k_long_base<3, false>), dim3(grid), dim3(256), 0, st, a);
  else if (a.twists) hipLaunchKernelGGL((k_long_base<2, false>), dim3(grid), dim3(256), 0, st, a);
  else if (a.J) hipLaunchKernelGGL((k_long_base<1, false>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_long_base<0, false>), dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t rdyn_launch_long_ext(int n_joints, const RdynKinExtArgs& a, hipStream_t st)
{
  (void)n_joints;
  if (a.n_samples <= 0) return hipSuccess;
  const dim3 grid((unsigned)((a.n_samples + 255) / 256)), grid64((unsigned)((a.n_samples + 63) / 64));
  if (a.wrench || a.tau)
  {
    // (the joint torques' tile behind the wrench ring: at most 64 (RDYN_MAX_JOINTS | 1) doubles)
    if (a.staged) hipLaunchKernelGGL((k_long_ext<true, true>), grid64, dim3(64), (size_t)RecordRing<48>::BYTES + (size_t)64 * (RDYN_MAX_JOINTS | 1) * 8, st, a);
    else hipLaunchKernelGGL((k_long_ext<true, false>), grid, dim3(256), 0, st, a);
  }
  else if (a.staged)
  {
    const int rings = (a.dtw_lin ? 1 : 0) + (a.dtw_nonlin ? 1 : 0) + (a.ddtw ? 1 : 0) + (a.ddtw_lin ? 1 : 0) + (a.ddtw_nonlin ? 1 : 0);
    hipLaunchKernelGGL((k_long_ext<false, true>), grid64, dim3(64), (size_t)rings * RecordRing<48>::BYTES, st, a);
  }
  else
    hipLaunchKernelGGL((k_long_ext<false, false>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}
