// rdyn_kin_ext.hip -- the split / jerk sweeps of rosdyn::Chain, base frame, as the
// reference states them (paths under /root/reference/rosdyn_core/include/rosdyn_core/internal/primitives_impl.h):
//   getDTwistLinearPart      :1029-1061   aL[l] = translate(aL[l-1], d) + S DDq
//   getDTwistNonLinearPart   :1063-1080   aN[l] = translate(aN[l-1], d) + (v x S) Dq
//   getDDTwist               :1185-1223   j[l]  = translate(j[l-1], d) + S DDDq + (v x S) DDq + (a x S + v x (v x S)) Dq
// One thread per sample; every output record is links x 6 doubles ([lin; ang] per link).
#include <hip/hip_runtime.h>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"

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

template <int NJ>
__global__ __launch_bounds__(256) void k_base_ext(const RdynKinExtArgs a)
{
  ChainPtr c = as_const(a.chain);
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.n_samples) return;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = a.dq ? a.dq + s * a.in_ss : nullptr;
  const double* __restrict__ ddqp = a.ddq ? a.ddq + s * a.in_ss : nullptr;
  const double* __restrict__ dddqp = a.dddq ? a.dddq + s * a.in_ss : nullptr;
  const int64_t es = a.out_se;
  auto put6 = [&](double* __restrict__ o, S6 x) {
    o[0] = x.l.x; o[es] = x.l.y; o[2 * es] = x.l.z; o[3 * es] = x.a.x; o[4 * es] = x.a.y; o[5 * es] = x.a.z;
  };
  const S6 zero = {mk(0, 0, 0), mk(0, 0, 0)};
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  S6 v = zero, acc = zero, aL = zero, aN = zero, jk = zero;
  if (a.dtw_lin) put6(a.dtw_lin + s * a.out_ss, zero);
  if (a.dtw_nonlin) put6(a.dtw_nonlin + s * a.out_ss, zero);
  if (a.ddtw) put6(a.ddtw + s * a.out_ss, zero);
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
      sincos(qf, &sn, &cs);
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
    aL = axpy6(shift(aL, d), S, ddqf);
    aN = axpy6(shift(aN, d), vxs, dqf);
    const S6 axs = xs(acc, S), vvxs = xs(v, vxs);
    S6 cq;
    cq.l = axs.l + vvxs.l;
    cq.a = axs.a + vvxs.a;
    jk = axpy6(axpy6(axpy6(shift(jk, d), S, dddqf), vxs, ddqf), cq, dqf);
    const int64_t off = (int64_t)(6 * (f + 1)) * es;
    if (a.dtw_lin) put6(a.dtw_lin + s * a.out_ss + off, aL);
    if (a.dtw_nonlin) put6(a.dtw_nonlin + s * a.out_ss + off, aN);
    if (a.ddtw) put6(a.ddtw + s * a.out_ss + off, jk);
  }
}

template <int NJ>
hipError_t launch_ext_nj(const RdynKinExtArgs& a, hipStream_t st)
{
  hipLaunchKernelGGL((k_base_ext<NJ>), dim3((unsigned)((a.n_samples + 255) / 256)), dim3(256), 0, st, a);
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
