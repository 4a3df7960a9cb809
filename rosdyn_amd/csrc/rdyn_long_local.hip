// rdyn_long_local.hip -- getRegressor / getJointInertia of chains with MORE INPUT JOINTS than the unrolled kernels sweep (11 ..
// RDYN_MAX_JOINTS input joints; chains of more than RDYN_MAX_SWEPT_JOINTS joints that have no reduced companion).
//
// The reference's default build has no bound on the number of joints (rosdyn_core/CMakeLists.txt:12-16); the kernels of
// rdyn_kernels.hip keep one unit twist per upstream joint in registers (instantiated per joint count, every link unrolled), which ends
// at ten.  Here the link loop AND the row loop are rolled (run-time trip counts, joint constants by scalar loads at wave-uniform
// offsets into RdynLongChainConst) and the per-joint state lives in wave-private LDS, element-major ([value][joint][lane]: a lane
// reads and writes its own column, conflict-free):
//
//   k_long_regressor   getRegressor (+ the fused joint torque tau = Y pi), primitives_impl.h:1295-1355.
//     One forward sweep in base-frame coordinates exactly as getTwist / getDTwist state it (:1007-1008, :1116-1117); joint l parks its
//     axis z_l and origin p_l (6 doubles) when the sweep passes it.  At link f the kinematic quantities are rotated into the link's own
//     frame (w, al, d = R'(a + w x v - g)) and row l <= f of the link's 10 columns is j_l' W'_f with the closed form of the reference's
//     ten basis-matrix products (:1324-1339, the same expression as rdyn_local_sweep_body.inc),
//         W'_f = [ d | [al]x + [w]x[w]x | 0 ;  0 | -[d]x | L(al) + [w]x L(w) ],
//     j_l = the unit twist of joint l at link f's origin in link f's axes: revolute (R' (z_l x (p_f - p_l)), R' z_l), prismatic
//     (R' z_l, 0) -- the transposed operator of :1341-1347.  Rows l > f are the structural zeros of :690-691, written explicitly.
//     ~70 fp64 fma per (row, link) and 10 stores; the output (n x 10 nJ doubles per sample: 32 KB at 20 joints) bounds the kernel.
//   k_long_inertia     getJointInertia, :1357-1379: M = sum_f J_f' I_f J_f evaluated by composite bodies -- M(l1, l2) = s_l1' Ic_l2 s_l2
//     for l1 <= l2, Ic_l = the spatial inertia of everything downstream of joint l.  A spatial inertia referred to the BASE origin and
//     axes is ten numbers (m, h = m c, I_O) that simply add, so Ic_l = total - (the links upstream of joint l): two forward passes,
//     no per-link storage (the device of rdyn_long_kin.hip's wrench recursion); the subtraction costs ~1e-16 of the chain's total
//     inertia, far inside the parity tolerance.  O(n^2) instead of the reference's O(n^3).
// Joint torques of such chains: the wrench recursion of rdyn_long_kin.hip (k_long_ext).
#include <hip/hip_runtime.h>
#include <atomic>
#include <type_traits>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"

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

// wave-private per-joint state: value v of joint j of the lane's sample at st[(v * nj + j) * 64 + lane]
struct JointState
{
  double* st;
  int nj;
  __device__ __forceinline__ double& at(int v, int j) const { return st[(v * nj + j) * 64]; }
  __device__ __forceinline__ void put3(int v0, int j, V3 x) const
  {
    at(v0, j) = x.x;
    at(v0 + 1, j) = x.y;
    at(v0 + 2, j) = x.z;
  }
  __device__ __forceinline__ V3 get3(int v0, int j) const { return mk(at(v0, j), at(v0 + 1, j), at(v0 + 2, j)); }
};

// columns 2 G, 2 G + 1 of one row of a link's block (the closed form above, one column group at a time: the staged kernel forms a link's
// block in five passes of two columns, see below)
struct RowCtx
{
  V3 w, al, dd;
  double b00, b01, b02, b10, b11, b12, b20, b21, b22;
};
template <int G>
__device__ __forceinline__ void y_pair(const RowCtx& k, V3 L, V3 A, double& ya, double& yb)
{
  if constexpr (G == 0)
  {
    const V3 dxA = cross(k.dd, A);
    ya = dot(L, k.dd);
    yb = fma(L.x, k.b00, fma(L.y, k.b10, fma(L.z, k.b20, dxA.x)));
  }
  else if constexpr (G == 1)
  {
    const V3 dxA = cross(k.dd, A);
    ya = fma(L.x, k.b01, fma(L.y, k.b11, fma(L.z, k.b21, dxA.y)));
    yb = fma(L.x, k.b02, fma(L.y, k.b12, fma(L.z, k.b22, dxA.z)));
  }
  else
  {
    const V3 x = cross(A, k.w);
    if constexpr (G == 2)
    {
      ya = fma(A.x, k.al.x, x.x * k.w.x);
      yb = fma(A.x, k.al.y, fma(A.y, k.al.x, fma(x.x, k.w.y, x.y * k.w.x)));
    }
    else if constexpr (G == 3)
    {
      ya = fma(A.x, k.al.z, fma(A.z, k.al.x, fma(x.x, k.w.z, x.z * k.w.x)));
      yb = fma(A.y, k.al.y, x.y * k.w.y);
    }
    else
    {
      ya = fma(A.y, k.al.z, fma(A.z, k.al.y, fma(x.y, k.w.z, x.z * k.w.y)));
      yb = fma(A.z, k.al.z, x.z * k.w.z);
    }
  }
}

typedef double ll_d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double ll_d2a __attribute__((ext_vector_type(2), aligned(16)));
__device__ __forceinline__ void ll_wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// STAGE 0: every value stored from the lane that computed it (element-major / free strides: 512 contiguous bytes per store instruction).
// STAGE 1 / 2: the ROW-CONTIGUOUS layouts -- per-sample images (stride_row 1, stride_col n: what a rosdyn::Chain caller receives,
// primitives_impl.h:1350-1354) and the stacked (N n) x P matrix (stride_sample n).  Stored from the computing lane these are 8-byte
// pieces on 64 different lines per store instruction (0.65 TB/s measured, profiles/r6/long_regressor.txt); here a link's block goes
// through a wave-private LDS tile two columns at a time and leaves 16 bytes per lane:
//   images   tile [sample][2 n + 1]: a sample's piece is ONE run of 16 n bytes of its image
//   stacked  tile [column][64 n + 2]: a column's 64 n values are ONE run of 512 n bytes of the matrix
// The row loop runs once per column group (the unit twist of a row is rebuilt from the parked axis and origin: ~24 fma -- cheaper
// than parking ten values per row and link), default-policy stores (the pieces of neighbouring column groups complete each other's
// lines in L2).
template <int STAGE>
__global__ __launch_bounds__(64) void k_long_regressor(const RdynLongLocalArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double joint_lds[];  // [7][nj][64]: z (3), p (3), tau (1); STAGE: + the tile
  LongChainPtr c = as_const_long(a.chain_long);
  const int nj = c->n_joints, n = c->n_active;
  const int lane = threadIdx.x;
  const int64_t s_wave = (int64_t)blockIdx.x * 64;
  const int64_t left = a.n_samples - s_wave;
  const int valid = left < 64 ? (int)left : 64;
  const bool live = lane < valid;
  if (STAGE == 0 && !live) return;
  const int64_t s = s_wave + (live ? lane : valid - 1);  // (staged: lanes past the batch repeat the last sample and take part in the copy-out)
  const JointState js = {joint_lds + lane, nj};
  double* const tile = joint_lds + 7 * nj * 64;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = a.dq ? a.dq + s * a.in_ss : nullptr;
  const double* __restrict__ ddqp = a.ddq ? a.ddq + s * a.in_ss : nullptr;
  double* const ys = a.Y + s * a.y_ss;
  double* const ywave = a.Y + s_wave * a.y_ss;
  const bool nt = a.y_ss == 1;  // element-major: every store instruction of the wave is 512 contiguous bytes, written once
  const V3 g = mk(c->g[0], c->g[1], c->g[2]);

  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3 p = mk(0, 0, 0);
  V3 vlin = mk(0, 0, 0), vang = mk(0, 0, 0), alin = mk(0, 0, 0), aang = mk(0, 0, 0);
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
      if (dqp) dqf = dqp[o];
      if (ddqp) ddqf = ddqp[o];
    }
    V3 zl, d;
    frame_step(J, qf, R, p, zl, d);
    {
      // twists (getTwist, :1007-1008) and spatial accelerations (getDTwist, :1116-1117), base frame, at the link's origin
      V3 Sl = mk(0, 0, 0), Sa = mk(0, 0, 0);
      if (type == RDYN_REVOLUTE) Sa = zl;
      else if (type == RDYN_PRISMATIC) Sl = zl;
      const V3 nvl = axpy(vlin + cross(vang, d), Sl, dqf);
      const V3 nva = axpy(vang, Sa, dqf);
      const V3 cl = cross(nva, Sl) + cross(nvl, Sa);  // spatialCrossProduct(v, S), sva.h:88-93
      const V3 ca = cross(nva, Sa);
      alin = axpy(axpy(alin + cross(aang, d), cl, dqf), Sl, ddqf);
      aang = axpy(axpy(aang, ca, dqf), Sa, ddqf);
      vlin = nvl;
      vang = nva;
    }
    js.put3(0, f, zl);
    js.put3(3, f, p);
    js.at(6, f) = 0.0;
    // ---- link f + 1 in its own frame: closed-form wrench regressor (rdyn_local_sweep_body.inc)
    RowCtx k;
    k.w = rotT(R, vang);
    k.al = rotT(R, aang);
    k.dd = rotT(R, alin - g) + cross(k.w, rotT(R, vlin));
    {
      const V3 w = k.w, al = k.al;
      const double wxy = w.x * w.y, wxz = w.x * w.z, wyz = w.y * w.z;
      const double wxx = w.x * w.x, wyy = w.y * w.y, wzz = w.z * w.z;
      k.b00 = -(wyy + wzz); k.b01 = wxy - al.z; k.b02 = wxz + al.y;
      k.b10 = wxy + al.z; k.b11 = -(wxx + wzz); k.b12 = wyz - al.x;
      k.b20 = wxz - al.y; k.b21 = wyz + al.x; k.b22 = -(wxx + wyy);
    }
    const RDYN_CONST_AS double* pi = J.pi;
    // the unit twist of joint l at this link's origin, this link's axes
    auto unit_twist = [&](JointRef Jl, int l, V3& L, V3& A) {
      const V3 z = js.get3(0, l);
      if (Jl.type == RDYN_REVOLUTE)
      {
        A = rotT(R, z);
        L = rotT(R, cross(z, p - js.get3(3, l)));
      }
      else
      {
        A = mk(0, 0, 0);
        L = Jl.type == RDYN_PRISMATIC ? rotT(R, z) : mk(0, 0, 0);
      }
    };
    if constexpr (STAGE == 0)
    {
      double* const yf = ys + (int64_t)(10 * f) * a.y_sc;
#pragma unroll 1
      for (int l = 0; l < nj; ++l)
      {
        JointRef Jl = c->j[l];
        const int row = Jl.in_idx;
        if (row < 0) continue;
        double* const yr = yf + row * a.y_sr;
        double y[10];
        if (l <= f)
        {
          V3 L, A;
          unit_twist(Jl, l, L, A);
          y_pair<0>(k, L, A, y[0], y[1]);
          y_pair<1>(k, L, A, y[2], y[3]);
          y_pair<2>(k, L, A, y[4], y[5]);
          y_pair<3>(k, L, A, y[6], y[7]);
          y_pair<4>(k, L, A, y[8], y[9]);
          double tl = js.at(6, l);
#pragma unroll
          for (int e = 0; e < 10; ++e) tl = fma(y[e], pi[e], tl);
          js.at(6, l) = tl;
        }
        else
        {
#pragma unroll
          for (int e = 0; e < 10; ++e) y[e] = 0.0;  // row of a joint downstream of this link (:690-691)
        }
        if (nt)
        {
#pragma unroll
          for (int e = 0; e < 10; ++e) __builtin_nontemporal_store(y[e], yr + e * a.y_sc);
        }
        else
        {
#pragma unroll
          for (int e = 0; e < 10; ++e) yr[e * a.y_sc] = y[e];
        }
      }
    }
    else
    {
      const int run = 2 * n, runp = run + 1, colp = 64 * n + 2;
      double* const mine = STAGE == 1 ? tile + lane * runp : tile + lane * n;
      const int pstep = STAGE == 1 ? n : colp;
      auto group = [&](auto gtag) {
        constexpr int G = decltype(gtag)::value;
#pragma unroll 1
        for (int l = 0; l < nj; ++l)
        {
          JointRef Jl = c->j[l];
          const int row = Jl.in_idx;
          if (row < 0) continue;
          double ya = 0.0, yb = 0.0;  // (l > f: row of a joint downstream of this link, :690-691)
          if (l <= f)
          {
            V3 L, A;
            unit_twist(Jl, l, L, A);
            y_pair<G>(k, L, A, ya, yb);
            js.at(6, l) = fma(yb, pi[2 * G + 1], fma(ya, pi[2 * G], js.at(6, l)));
          }
          mine[row] = ya;
          mine[pstep + row] = yb;
        }
        ll_wave_fence();
        if (STAGE == 1)
        {
          // lps lanes per sample (a power of two >= the piece's n 16-byte chunks), 64 / lps samples per store instruction
          const int lps = n <= 16 ? 16 : 32, spi = 64 / lps;
          const int wch = lane & (lps - 1), sl = lane / lps;
          const double* src = tile + sl * runp + 2 * wch;
          double* dst = ywave + (int64_t)sl * a.y_ss + (int64_t)(10 * f + 2 * G) * a.y_sc + 2 * wch;
#pragma unroll 4
          for (int it = 0; it < lps; ++it)
          {
            if (wch < n && it * spi + sl < valid) *(ll_d2a*)dst = (ll_d2a)(*(const ll_d2u*)src);
            src += spi * runp;
            dst += (int64_t)spi * a.y_ss;
          }
        }
        else
        {
#pragma unroll
          for (int pp = 0; pp < 2; ++pp)
          {
            double* const yc = ywave + (int64_t)(10 * f + 2 * G + pp) * a.y_sc;
#pragma unroll 2
            for (int it = 0; it < (n + 1) / 2; ++it)
            {
              const int ch = it * 64 + lane;  // 16-byte piece of the column's run: values 2 ch, 2 ch + 1 of 64 n
              if (ch < 32 * n)
              {
                const ll_d2a v = *(const ll_d2a*)(tile + pp * colp + 2 * ch);
                if (2 * ch + 1 < valid * n) *(ll_d2a*)(yc + 2 * ch) = v;
                else if (2 * ch < valid * n) yc[2 * ch] = v.x;
              }
            }
          }
        }
        ll_wave_fence();
      };
      group(std::integral_constant<int, 0>());
      group(std::integral_constant<int, 1>());
      group(std::integral_constant<int, 2>());
      group(std::integral_constant<int, 3>());
      group(std::integral_constant<int, 4>());
    }
  }
  if (a.bcol && live)
  {
    const double* __restrict__ bp = a.bcol + s * a.in_ss;
    double* const yb = ys + (int64_t)a.bcol_col * a.y_sc;
#pragma unroll 1
    for (int l = 0; l < nj; ++l)
    {
      const int row = c->j[l].in_idx;
      if (row >= 0) yb[row * a.y_sr] = bp[row * a.in_sj];
    }
  }
  if (a.tau && live)
  {
    double* __restrict__ tp = a.tau + s * a.tau_ss;
#pragma unroll 1
    for (int l = 0; l < nj; ++l)
    {
      const int row = c->j[l].in_idx;
      if (row >= 0) tp[row * a.tau_sj] = js.at(6, l);
    }
  }
}

// ten numbers of a spatial inertia about the BASE origin, base axes: m, h = m c, I_O (xx xy xz yy yz zz)
struct Inertia10
{
  double m;
  V3 h;
  double I[6];
};
__device__ __forceinline__ Inertia10 zero10()
{
  Inertia10 r;
  r.m = 0.0;
  r.h = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 6; ++i) r.I[i] = 0.0;
  return r;
}
// the link's [m, m c, Io] (its own frame, about its own origin: primitives_impl.h:399-417) referred to the base origin and axes:
//   h_b = m p + R h,   I_O = R Io R' + (m |p|^2 + 2 p.hb) 1 - (m p p' + p hb' + hb p'),   hb = R h
template <class Ptr>
__device__ __forceinline__ Inertia10 to_base(Ptr pi, const double (&R)[9], V3 p)
{
  Inertia10 r;
  const double m = pi[0];
  const V3 hb = rot(R, mk(pi[1], pi[2], pi[3]));
  r.m = m;
  r.h = axpy(hb, p, m);
  // R Io R': columns of Io R' first
  const V3 c0 = symv(pi + 4, mk(R[0], R[1], R[2])), c1 = symv(pi + 4, mk(R[3], R[4], R[5])), c2 = symv(pi + 4, mk(R[6], R[7], R[8]));  // Io r_i (r_i = row i of R)
  const V3 r0 = mk(R[0], R[1], R[2]), r1 = mk(R[3], R[4], R[5]), r2 = mk(R[6], R[7], R[8]);
  const double tr = m * dot(p, p) + 2.0 * dot(p, hb);
  r.I[0] = dot(r0, c0) + tr - (m * p.x * p.x + 2.0 * p.x * hb.x);
  r.I[1] = dot(r0, c1) - (m * p.x * p.y + p.x * hb.y + hb.x * p.y);
  r.I[2] = dot(r0, c2) - (m * p.x * p.z + p.x * hb.z + hb.x * p.z);
  r.I[3] = dot(r1, c1) + tr - (m * p.y * p.y + 2.0 * p.y * hb.y);
  r.I[4] = dot(r1, c2) - (m * p.y * p.z + p.y * hb.z + hb.y * p.z);
  r.I[5] = dot(r2, c2) + tr - (m * p.z * p.z + 2.0 * p.z * hb.z);
  return r;
}

__global__ __launch_bounds__(64) void k_long_inertia(const RdynLongLocalArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double joint_lds[];  // [6][nj][64]: the unit twist of joint j about the base origin (lin, ang)
  LongChainPtr c = as_const_long(a.chain_long);
  const int nj = c->n_joints, n = c->n_active;
  const int lane = threadIdx.x;
  const int64_t s = (int64_t)blockIdx.x * 64 + lane;
  if (s >= a.n_samples) return;
  const JointState js = {joint_lds + lane, nj};
  const double* __restrict__ qp = a.q + s * a.in_ss;
  double* __restrict__ mp = a.M + s * a.m_ss;
  Inertia10 total = zero10(), upstream = zero10();
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass)
  {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3 p = mk(0, 0, 0);
#pragma unroll 1
    for (int f = 0; f < nj; ++f)
    {
      JointRef J = c->j[f];
      const int idx = J.in_idx;
      V3 zl, d;
      frame_step(J, idx >= 0 ? qp[idx * a.in_sj] : 0.0, R, p, zl, d);
      const Inertia10 own = to_base(J.pi, R, p);
      if (pass == 0)
      {
        total.m += own.m;
        total.h = total.h + own.h;
#pragma unroll
        for (int i = 0; i < 6; ++i) total.I[i] += own.I[i];
        // unit twist of joint f referred to the base origin: revolute (p x z, z), prismatic (z, 0)
        V3 sl = mk(0, 0, 0), sa = mk(0, 0, 0);
        if (J.type == RDYN_REVOLUTE)
        {
          sl = cross(p, zl);
          sa = zl;
        }
        else if (J.type == RDYN_PRISMATIC)
          sl = zl;
        js.put3(0, f, sl);
        js.put3(3, f, sa);
        continue;
      }
      if (idx >= 0)
      {
        // composite body downstream of joint f: everything but the links upstream of it
        const double m = total.m - upstream.m;
        const V3 h = total.h - upstream.h;
        double I[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) I[i] = total.I[i] - upstream.I[i];
        const V3 sl = js.get3(0, f), sa = js.get3(3, f);
        // momentum of the composite body under the joint's unit twist: F = m v + w x h, N = h x v + I_O w
        const V3 F = axpy(cross(sa, h), sl, m);
        const V3 N = cross(h, sl) + symv(I, sa);
#pragma unroll 1
        for (int l = 0; l <= f; ++l)
        {
          const int r1 = c->j[l].in_idx;
          if (r1 < 0) continue;
          const double v = dot(js.get3(0, l), F) + dot(js.get3(3, l), N);
          mp[(int64_t)(idx * n + r1) * a.m_se] = v;
          mp[(int64_t)(r1 * n + idx) * a.m_se] = v;
        }
      }
      upstream.m += own.m;
      upstream.h = upstream.h + own.h;
#pragma unroll
      for (int i = 0; i < 6; ++i) upstream.I[i] += own.I[i];
    }
  }
}

// more than 64 KB of dynamic LDS needs the attribute, once per kernel and device (slot: 0 inertia, 1..3 the regressor's STAGE 0..2)
hipError_t allow_big_lds(const void* fn, size_t bytes, int slot)
{
  if (bytes <= 64 * 1024) return hipSuccess;
  static std::atomic<uint64_t> done[4];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done[slot].load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done[slot].fetch_or(bit, std::memory_order_release);
  return e;
}
}  // namespace

size_t rdyn_long_local_lds_bytes(int mode, int n_joints) { return (size_t)(mode == RDYN_MODE_INERTIA ? 6 : 7) * n_joints * 64 * sizeof(double); }

hipError_t rdyn_launch_long_local(int mode, int n_joints, const RdynLongLocalArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  size_t lds = rdyn_long_local_lds_bytes(mode, n_joints);
  const dim3 grid((unsigned)((a.n_samples + 63) / 64));
  if (mode == RDYN_MODE_INERTIA)
  {
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = allow_big_lds((const void*)k_long_inertia, lds, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_long_inertia, grid, dim3(64), lds, st, a);
    return hipGetLastError();
  }
  // a.stage (decided by the host: a row-contiguous layout, 16-byte aligned Y, even strides): the tile behind the joint state
  const size_t tile = a.stage == 1 ? (size_t)64 * (2 * a.n_active + 1) * 8 : (a.stage == 2 ? (size_t)2 * (64 * a.n_active + 2) * 8 : 0);
  const int stage = (a.stage && lds + tile <= 160 * 1024) ? a.stage : 0;
  if (stage) lds += tile;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const void* fn = stage == 1 ? (const void*)k_long_regressor<1> : (stage == 2 ? (const void*)k_long_regressor<2> : (const void*)k_long_regressor<0>);
  hipError_t e = allow_big_lds(fn, lds, 1 + stage);
  if (e != hipSuccess) return e;
  if (stage == 1) hipLaunchKernelGGL(k_long_regressor<1>, grid, dim3(64), lds, st, a);
  else if (stage == 2) hipLaunchKernelGGL(k_long_regressor<2>, grid, dim3(64), lds, st, a);
  else hipLaunchKernelGGL(k_long_regressor<0>, grid, dim3(64), lds, st, a);
  return hipGetLastError();
}
