// rdyn_rowpair.hip -- regressor kernel for row-contiguous output layouts (see the comment below).
#include <hip/hip_runtime.h>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"

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

// ---------------------------------------------------------------------------------------------------
// k_rowpair_sweep<NJ> -- regressor (+ fused torque) for ROW-CONTIGUOUS output layouts (stride_row == 1: the
// stacked column-major (N*n) x P matrix and the per-sample Eigen image), sample-major inputs.
//
// With one thread per sample such layouts make every lane write 8 bytes at a stride of n (or n*P) doubles:
// 64 partial-line requests per store instruction (measured 3.5-3.9 ms per 1e6 evaluations, 6x the
// element-major time).  Here G = ceil(n / 2) consecutive lanes share one sample; lane k owns rows 2k, 2k+1 and
// stores them as ONE 16-byte element [Y(2k, p), Y(2k+1, p)].  In the stacked layout consecutive lanes then write
// consecutive 16 bytes: every store instruction of a wave is 1 KiB contiguous.  Each lane repeats the (cheap,
// ~1.1 k fp64 op) forward sweep of its sample but carries only its two rows' joint twists j_l (zero until the
// row's joint is reached, so the same code path serves "not yet active" and structural zeros without branches).
// Arithmetic per sample grows ~2x and stays well below the HBM time.
namespace
{

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ uint32_t div_small(uint32_t t, int G)
{
  switch (G)  // wave-uniform
  {
  case 1: return t;
  case 2: return t >> 1;
  case 3: return (uint32_t)(((uint64_t)t * 0xAAAAAAABull) >> 33);
  case 4: return t >> 2;
  default: return (uint32_t)(((uint64_t)t * 0xCCCCCCCDull) >> 34);  // 5
  }
}

// NT: nontemporal Y stores.  Right for the stacked layout (every wave store is 1 KiB of full lines, never re-read:
// +11 % measured); wrong for the per-sample image, whose 48-byte runs rely on L2 write-combining (3x slower with nt).
template <int NJ, bool NT>
__global__ __launch_bounds__(256) void k_rowpair_sweep(const RdynSweepArgs a, const int G)
{
  ChainPtr c = as_const(a.chain);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t s = div_small(t, G);
  if ((int64_t)s >= a.n_samples) return;
  const int k = (int)(t - s * (uint32_t)G);
  const int n = c->n_active;
  const int r0 = 2 * k, r1 = 2 * k + 1;  // r1 == n for the last pair of an odd n: that half is not stored

  // the regressor needs q, Dq and DDq (checked by the API): no null tests, no per-lane branches
  const double* __restrict__ qp = a.q + (int64_t)s * a.in_ss;
  const double* __restrict__ dqp = a.dq + (int64_t)s * a.in_ss;
  const double* __restrict__ ddqp = a.ddq + (int64_t)s * a.in_ss;
  // stride_row == 1.  Per-lane part of the address as a 32-bit byte offset (the API splits launches so that it fits),
  // wave-uniform part in SGPRs: keeps ten 64-bit pointers per link out of the VGPR file.
  const uint32_t yv = (uint32_t)(((int64_t)s * a.y_ss + r0) * 8);

  V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
  V3 acc = mk(-c->g[0], -c->g[1], -c->g[2]);
  V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);
  double tau0 = 0.0, tau1 = 0.0;

  // No per-link arrays here.  Stacked layout (NT): a ROLLED loop keeps the live set to one link (137 VGPRs, 3 waves per SIMD;
  // fully unrolled it needs > 256) and measured 527 vs 539 us; the per-sample image measured better unrolled (671 vs 810 us).
#pragma unroll(NT ? 1 : NJ)
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
      dqf = dqp[o];
      ddqf = ddqp[o];
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
      const V3 an = rotT(R, acc + cross(al, tt));
      w = wn; vl = vn; al = aln; acc = an;
      const V3 nL0 = rotT(R, L0 + cross(A0, tt));
      A0 = rotT(R, A0);
      L0 = nL0;
      const V3 nL1 = rotT(R, L1 + cross(A1, tt));
      A1 = rotT(R, A1);
      L1 = nL1;
    }
    const V3 u = ld3(J.u);
    V3 sl = mk(0, 0, 0), sa = mk(0, 0, 0);  // this joint's own unit twist in its child frame
    if (type == RDYN_REVOLUTE)
    {
      acc = axpy(acc, cross(vl, u), dqf);
      al = axpy(axpy(al, cross(w, u), dqf), u, ddqf);
      w = axpy(w, u, dqf);
      sa = u;
    }
    else if (type == RDYN_PRISMATIC)
    {
      acc = axpy(axpy(acc, cross(w, u), dqf), u, ddqf);
      vl = axpy(vl, u, dqf);
      sl = u;
    }
    if (idx >= 0)
    {
      const bool m0 = (idx == r0), m1 = (idx == r1);  // per-lane: does one of my rows start at this joint?
      // component-wise selects: a struct-typed `m ? a : b` is lowered through memory and lands in scratch
      L0 = mk(m0 ? sl.x : L0.x, m0 ? sl.y : L0.y, m0 ? sl.z : L0.z);
      A0 = mk(m0 ? sa.x : A0.x, m0 ? sa.y : A0.y, m0 ? sa.z : A0.z);
      L1 = mk(m1 ? sl.x : L1.x, m1 ? sl.y : L1.y, m1 ? sl.z : L1.z);
      A1 = mk(m1 ? sa.x : A1.x, m1 ? sa.y : A1.y, m1 ? sa.z : A1.z);
    }

    const V3 d = acc + cross(w, vl);
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
    const RDYN_CONST_AS double* pi = J.pi;
    char* const yc = (char*)(a.Y + (int64_t)(10 * f) * a.y_sc);
    const bool pair = r1 < n;
#pragma unroll
    for (int p = 0; p < 10; ++p)
    {
      tau0 = fma(y0[p], pi[p], tau0);
      tau1 = fma(y1[p], pi[p], tau1);
      double* __restrict__ dst = (double*)(yc + p * (a.y_sc * 8) + yv);
      if (pair)
      {
        d2u v = {y0[p], y1[p]};
        if (NT) __builtin_nontemporal_store(v, (d2u*)dst);
        else *(d2u*)dst = v;
      }
      else if (NT)
        __builtin_nontemporal_store(y0[p], dst);
      else
        dst[0] = y0[p];
    }
  }
  if (a.tau)
  {
    double* __restrict__ tp = a.tau + (int64_t)s * a.tau_ss + r0;  // tau_sj == 1 (sample-major)
    if (r1 < n)
    {
      d2u v = {tau0, tau1};
      *(d2u*)tp = v;
    }
    else
      tp[0] = tau0;
  }
}

template <int NJ>
hipError_t launch_rowpair_nj(const RdynSweepArgs& a, int G, hipStream_t st)
{
  const int64_t threads = a.n_samples * G;
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (a.y_ss == 2 * G || a.y_ss == 2 * G - 1)  // stacked layout: stride_sample == n -> consecutive lanes are contiguous
    hipLaunchKernelGGL((k_rowpair_sweep<NJ, true>), grid, dim3(256), 0, st, a, G);
  else
    hipLaunchKernelGGL((k_rowpair_sweep<NJ, false>), grid, dim3(256), 0, st, a, G);
  return hipGetLastError();
}
}  // namespace

// requires: y_sr == 1, tau_sj == 1 (or tau null), n_active <= 10, n_samples * ceil(n/2) < 2^32
hipError_t rdyn_launch_rowpair_sweep(int n_joints, int n_active, const RdynSweepArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  const int G = (n_active + 1) / 2;
#define CALL(N) launch_rowpair_nj<N>(a, G, st)
  RDYN_DISPATCH_NJ(n_joints, CALL)
#undef CALL
}
