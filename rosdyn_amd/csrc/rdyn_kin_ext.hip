// rdyn_kin_ext.hip -- the split / jerk sweeps of rosdyn::Chain, base frame, as the
// reference states them (paths under /root/reference/rosdyn_core/include/rosdyn_core/internal/primitives_impl.h):
//   getDTwistLinearPart      :1029-1061   aL[l] = translate(aL[l-1], d) + S DDq
//   getDTwistNonLinearPart   :1063-1080   aN[l] = translate(aN[l-1], d) + (v x S) Dq
//   getDDTwist               :1185-1223   j[l]  = translate(j[l-1], d) + S DDDq + (v x S) DDq + (a x S + v x (v x S)) Dq
//   getDDTwistLinearPart     :1126-1154   jL[l] = translate(jL[l-1], d) + S DDDq
//   getDDTwistNonLinearPart  :1156-1183   jN[l] = translate(jN[l-1], d) + (v x S) DDq + (a x S + v x (v x S)) Dq
//   getWrench                :1225-1262   w[l]  = T(-ext[l]) + inertial[l] + gravity[l] + dualTranslate(w[l+1], p_l - p_l+1)
// One thread per sample; every output record is links x 6 doubles ([lin; ang] per link).
// The wrench recursion runs tool -> base in the reference; here the forward sweep parks every link's OWN wrench, referred to the
// base origin (dualTranslate is additive in the offset), in wave-private LDS, and a short backward pass over the thread's own
// records forms the suffix sums: no per-link frame storage, no 6 (L) accumulators in registers.
#include <hip/hip_runtime.h>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_record_stage.h"

namespace
{
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

#ifndef RDYN_KIN_EXT_WAVES
#define RDYN_KIN_EXT_WAVES 1  // minimum waves per SIMD asked of the register allocator.  3 and 4 were measured: hipcc then spills (60-404 B of scratch) and getWrench / getDDTwist slow down 1.15-5x (profiles/r2/perf_sheet.txt)
#endif
// WRENCH: 64-thread workgroups (the wave parks every link's own wrench in LDS, 48 (NJ + 1) bytes per thread; no barrier)
// STAGED (a.staged: sample-major records at their natural stride, line-aligned outputs): 64-thread workgroups; the records of a FULL
// wave leave through wave-private LDS in whole lines (rdyn_record_stage.h) -- one ring per requested split / jerk output, the wrench
// records from the tile they are parked in anyway (then laid out [sample][6 (NJ + 1) | 1])
// MASK (split / jerk sweeps): which of the five outputs this instantiation can produce -- bit 0 dtw_lin, 1 dtw_nonlin, 2 ddtw, 3 ddtw_lin,
// 4 ddtw_nonlin; 31 = any combination.  The reference's getters ask for ONE at a time (primitives.h:468-488) and so do the facade and the
// Python mirror: with the other four recursions compiled out a call carries a fraction of the 84 registers of running state (203 for
// the all-outputs kernel: two waves per SIMD on a kernel that is bound by the waves in flight).
template <int NJ, bool WRENCH, bool STAGED, int MASK = 31>
__global__ __launch_bounds__((WRENCH || STAGED) ? 64 : 256, RDYN_KIN_EXT_WAVES) void k_base_ext(const RdynKinExtArgs a)
{
  constexpr bool W0 = (MASK & 1) != 0, W1 = (MASK & 2) != 0, W2 = (MASK & 4) != 0, W3 = (MASK & 8) != 0, W4 = (MASK & 16) != 0;
  constexpr int BS = (WRENCH || STAGED) ? 64 : 256;
  extern __shared__ __attribute__((aligned(16))) double own_lds[];  // WRENCH: [6 (NJ + 1)][64]; STAGED: the rings / the record tile
  ChainPtr c = as_const(a.chain);
  const int64_t s = (int64_t)blockIdx.x * BS + threadIdx.x;
  const bool stg = STAGED && a.n_samples - (int64_t)blockIdx.x * BS >= 64;  // wave-uniform
  RecordRing<48> rings[(STAGED && !WRENCH) ? 5 : 1];
  SmallRecords wtile;
  if constexpr (STAGED)
  {
    if (stg)
    {
      if constexpr (WRENCH)
      {
        wtile.init((char*)own_lds, 6 * (NJ + 1), threadIdx.x);
        // the external wrenches arrive in the tile the link wrenches leave from: ext_of(link) reads the lane's own slot before park6(link)
        // overwrites it (a.ext_staged: natural stride, 16-byte aligned -- decided by the host)
        if (a.ext && a.ext_staged)
        {
          load_records_into_tile<3 * (NJ + 1)>(wtile.tile, wtile.prec, a.ext + (int64_t)blockIdx.x * BS * a.ext_ss, 6 * (NJ + 1), threadIdx.x);
          rs_wave_fence();
        }
      }
      else
      {
        char* lp = (char*)own_lds;
        double* const outs[5] = {W0 ? a.dtw_lin : nullptr, W1 ? a.dtw_nonlin : nullptr, W2 ? a.ddtw : nullptr, W3 ? a.ddtw_lin : nullptr,
                                 W4 ? a.ddtw_nonlin : nullptr};
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if (outs[k])
          {
            rings[k].init(lp, outs[k] + (int64_t)blockIdx.x * BS * a.out_ss, 48u * (NJ + 1), threadIdx.x);
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
  // output addressing = wave-uniform 64-bit base (SGPRs) + one 32-bit per-lane byte offset (the saddr store form): with a full
  // 64-bit pointer per record element hipcc keeps dozens of address pairs alive in the unrolled link loop (204 VGPRs)
  const int64_t blk_off = (int64_t)blockIdx.x * BS * a.out_ss;
  const uint32_t lane_off = threadIdx.x * (uint32_t)a.out_ss * 8u;
  auto put6 = [&](double* base, int64_t first_elem, S6 x) {
    char* const o = (char*)(base + blk_off + first_elem * es);  // uniform
    const int64_t eb = es * 8;
    *(double*)(o + lane_off) = x.l.x;
    *(double*)(o + eb + lane_off) = x.l.y;
    *(double*)(o + 2 * eb + lane_off) = x.l.z;
    *(double*)(o + 3 * eb + lane_off) = x.a.x;
    *(double*)(o + 4 * eb + lane_off) = x.a.y;
    *(double*)(o + 5 * eb + lane_off) = x.a.z;
  };
  const S6 zero = {mk(0, 0, 0), mk(0, 0, 0)};
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  S6 v = zero, acc = zero, aL = zero, aN = zero, jk = zero, jL = zero, jN = zero;
  // link `link` of split / jerk output k
  auto out6 = [&](const int k, double* base, const int link, S6 x) {
    if (!base) return;
    if (STAGED && stg)
    {
      const RecordRing<48>& ring = rings[(STAGED && !WRENCH) ? k : 0];
      ring.put((uint32_t)(48 * link), x.l.x);
      ring.put((uint32_t)(48 * link + 8), x.l.y);
      ring.put((uint32_t)(48 * link + 16), x.l.z);
      ring.put((uint32_t)(48 * link + 24), x.a.x);
      ring.put((uint32_t)(48 * link + 32), x.a.y);
      ring.put((uint32_t)(48 * link + 40), x.a.z);
      ring.flush(48 * link, 48 * (link + 1));
    }
    else
      put6(base, 6 * link, x);
  };
  if (!WRENCH)
  {
    if (W0) out6(0, a.dtw_lin, 0, zero);
    if (W1) out6(1, a.dtw_nonlin, 0, zero);
    if (W2) out6(2, a.ddtw, 0, zero);
    if (W3) out6(3, a.ddtw_lin, 0, zero);
    if (W4) out6(4, a.ddtw_nonlin, 0, zero);
  }
  // WRENCH: origins of links 0 .. NJ.  The per-link wrench accumulators do NOT live in registers (6 (NJ + 1) doubles pushed the
  // kernel to 200 VGPRs = 2 waves per SIMD on a streaming kernel): every link's OWN wrench is parked in LDS referred to the base
  // origin (moment + p x force), and a short backward pass over the thread's own records forms the suffix sums and refers each
  // one back to its link's origin.  (Round 2's first version parked them in the output records: the store -> load round trip
  // through L2 / HBM, at two waves per SIMD, left the kernel waiting on memory for two thirds of its cycles.)
  V3 po[WRENCH ? NJ + 1 : 1];
  // (staged: the tile the records are copied out of -- [sample][6 (NJ + 1) | 1], an odd pitch: conflict-free for the lanes' own records
  // and for the copy-out, which walks along a record)
  auto park6 = [&](int link, S6 x) {
    if (STAGED && stg)
    {
      double* const o = wtile.mine + 6 * link;
      o[0] = x.l.x; o[1] = x.l.y; o[2] = x.l.z; o[3] = x.a.x; o[4] = x.a.y; o[5] = x.a.z;
      return;
    }
    double* const o = own_lds + (6 * link) * 64 + threadIdx.x;
    o[0] = x.l.x; o[64] = x.l.y; o[128] = x.l.z; o[192] = x.a.x; o[256] = x.a.y; o[320] = x.a.z;
  };
  auto parked6 = [&](int link) -> S6 {
    S6 x;
    if (STAGED && stg)
    {
      const double* const o = wtile.mine + 6 * link;
      x.l = mk(o[0], o[1], o[2]);
      x.a = mk(o[3], o[4], o[5]);
      return x;
    }
    const double* const o = own_lds + (6 * link) * 64 + threadIdx.x;
    x.l = mk(o[0], o[64], o[128]);
    x.a = mk(o[192], o[256], o[320]);
    return x;
  };
  V3 p = mk(0, 0, 0);
  const V3 grav = mk(c->g[0], c->g[1], c->g[2]);
  auto ext_of = [&](int link) -> S6 {  // -ext_wrenches_in_link_frame.at(link), :1255
    S6 e = zero;
    if (STAGED && stg && a.ext && a.ext_staged)
    {
      const double* const o = wtile.mine + 6 * link;
      e.l = mk(-o[0], -o[1], -o[2]);
      e.a = mk(-o[3], -o[4], -o[5]);
    }
    else if (a.ext)
    {
      const char* const ep = (const char*)(a.ext + (int64_t)blockIdx.x * BS * a.ext_ss + (int64_t)(6 * link) * a.ext_se);  // uniform
      const uint32_t ev = threadIdx.x * (uint32_t)a.ext_ss * 8u;
      const int64_t eb = a.ext_se * 8;
      e.l = mk(-*(const double*)(ep + ev), -*(const double*)(ep + eb + ev), -*(const double*)(ep + 2 * eb + ev));
      e.a = mk(-*(const double*)(ep + 3 * eb + ev), -*(const double*)(ep + 4 * eb + ev), -*(const double*)(ep + 5 * eb + ev));
    }
    return e;
  };
  if (WRENCH)
  {
    po[0] = p;
    park6(0, ext_of(0));  // spatialTranformation(-ext, T_bl[0] = identity); no inertial / gravity term on the base link (:1233-1237)
  }
#pragma unroll
  for (int f = 0; f < NJ; ++f)
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
    const V3 zl = rot(R, ld3(J.up));
    const V3 d = rot(R, t);
    double Rn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Rn[r * 3 + cc] = fma(R[r * 3 + 0], Rpc[cc], fma(R[r * 3 + 1], Rpc[3 + cc], R[r * 3 + 2] * Rpc[6 + cc]));
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    S6 S = zero;
    if (type == RDYN_REVOLUTE) S.a = zl;
    else if (type == RDYN_PRISMATIC) S.l = zl;
    v = axpy6(shift(v, d), S, dqf);
    const S6 vxs = xs(v, S);
    acc = axpy6(axpy6(shift(acc, d), vxs, dqf), S, ddqf);   // getDTwist, :1116-1117 (needed by the jerk)
    if (!WRENCH)  // the wrench instantiation is launched on its own (rdyn_wrench): no split / jerk state in its registers
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
      if (W0) out6(0, a.dtw_lin, f + 1, aL);
      if (W1) out6(1, a.dtw_nonlin, f + 1, aN);
      if (W2) out6(2, a.ddtw, f + 1, jk);
      if (W3) out6(3, a.ddtw_lin, f + 1, jL);
      if (W4) out6(4, a.ddtw_nonlin, f + 1, jN);
    }
    if (WRENCH)
    {
      p = p + d;
      po[f + 1] = p;
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
      own.a = own.a + cross(p, own.l);  // referred to the base origin: suffix sums need no per-pair translation
      park6(f + 1, own);
    }
  }
  if (WRENCH)
  {
    // w[l] = sum over links f >= l, referred to link l's origin: spatialDualTranslation(w, p_l - p_f), :1255 (ang += lin x d)
    S6 run = zero;
#pragma unroll
    for (int l = NJ; l >= 0; --l)
    {
      const S6 own = parked6(l);
      run.l = run.l + own.l;
      run.a = run.a + own.a;
      S6 w;
      w.l = run.l;
      w.a = run.a - cross(po[l], run.l);
      if (STAGED && stg) park6(l, w);  // (the lane's own slot: read above, never again)
      else put6(a.wrench, 6 * l, w);
    }
    if (STAGED && stg) wtile.copy_out(a.wrench + (int64_t)blockIdx.x * BS * a.out_ss, threadIdx.x);
  }
  else if (STAGED && stg)
  {
    double* const outs[5] = {W0 ? a.dtw_lin : nullptr, W1 ? a.dtw_nonlin : nullptr, W2 ? a.ddtw : nullptr, W3 ? a.ddtw_lin : nullptr,
                             W4 ? a.ddtw_nonlin : nullptr};
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (outs[k]) rings[(STAGED && !WRENCH) ? k : 0].finish();
  }
}

template <int NJ>
hipError_t launch_ext_nj(const RdynKinExtArgs& a, hipStream_t st)
{
  const dim3 grid((unsigned)((a.n_samples + 255) / 256)), grid64((unsigned)((a.n_samples + 63) / 64));
  if (a.wrench)
  {
#ifndef RDYN_WRENCH_LDS_PAD
#define RDYN_WRENCH_LDS_PAD 0  // (timing experiment: fewer waves per CU through the LDS request)
#endif
    if (a.staged) hipLaunchKernelGGL((k_base_ext<NJ, true, true>), grid64, dim3(64), (size_t)((6 * (NJ + 1)) | 1) * 64 * sizeof(double) + RDYN_WRENCH_LDS_PAD, st, a);
    else hipLaunchKernelGGL((k_base_ext<NJ, true, false>), grid64, dim3(64), (size_t)6 * (NJ + 1) * 64 * sizeof(double), st, a);
  }
  else
  {
    const int mask = (a.dtw_lin ? 1 : 0) | (a.dtw_nonlin ? 2 : 0) | (a.ddtw ? 4 : 0) | (a.ddtw_lin ? 8 : 0) | (a.ddtw_nonlin ? 16 : 0);
    const int rings = __builtin_popcount((unsigned)mask);
    const size_t lds = (size_t)rings * RecordRing<48>::BYTES;
#define RDYN_EXT_LAUNCH(M_)                                                                                       \
  do                                                                                                              \
  {                                                                                                               \
    if (a.staged) hipLaunchKernelGGL((k_base_ext<NJ, false, true, M_>), grid64, dim3(64), lds, st, a);           \
    else hipLaunchKernelGGL((k_base_ext<NJ, false, false, M_>), grid, dim3(256), 0, st, a);                      \
  } while (0)
    switch (mask)  // one output: its own instantiation
    {
    case 1: RDYN_EXT_LAUNCH(1); break;
    case 2: RDYN_EXT_LAUNCH(2); break;
    case 4: RDYN_EXT_LAUNCH(4); break;
    case 8: RDYN_EXT_LAUNCH(8); break;
    case 16: RDYN_EXT_LAUNCH(16); break;
    default: RDYN_EXT_LAUNCH(31); break;
    }
#undef RDYN_EXT_LAUNCH
  }
  return hipGetLastError();
}
}  // namespace

hipError_t rdyn_launch_base_ext(int n_joints, const RdynKinExtArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  switch (n_joints)
  {
  case 1: return launch_ext_nj<1>(a, st);
  case 2: return launch_ext_nj<2>(a, st);
  case 3: return launch_ext_nj<3>(a, st);
  case 4: return launch_ext_nj<4>(a, st);
  case 5: return launch_ext_nj<5>(a, st);
  case 6: return launch_ext_nj<6>(a, st);
  case 7: return launch_ext_nj<7>(a, st);
  case 8: return launch_ext_nj<8>(a, st);
  case 9: return launch_ext_nj<9>(a, st);
  case 10: return launch_ext_nj<10>(a, st);
  default: return hipErrorInvalidValue;
  }
}
