// rdyn_ik.hip -- batched local inverse kinematics for gfx950.
//
// Reference (one pose per call, wall-clock bounded loop):
//   Chain::computeLocalIk          primitives_impl.h:1398-1433
//   Chain::computeWeigthedLocalIk  primitives_impl.h:1436-1468
//   getFrameDistance               frame_distance.h:44-49
// Every iteration: frames -> pose error e (6) -> if |w o e| < toll done -> tool Jacobian -> H = J' W J, f = -J' W e ->
//   dq = argmin 1/2 dq'H dq + f'dq   s.t.  q_min <= sol + dq <= q_max   (Eigen::solve_quadprog, Goldfarb-Idnani) -> sol += dq.
//
// Here: one thread per pose, one WAVE per workgroup (poses need different numbers of iterations; a 64-lane workgroup
// retires as soon as its slowest pose does).  Everything lives in registers: all loops over joints are unrolled over
// the template parameter NJ (chain joints), the QP is solved in CHAIN-joint coordinates with the joints that are fixed
// or not in the input list held at dq = 0, so every array index is a compile-time constant.
//
// The QP has only bound constraints, so the dual active-set method of Goldfarb & Idnani specialises to:
//   active set = variables held at their lower / upper bound (two bit masks per lane),
//   H-operator   z = H n+   : solve H_FF z_F = sigma e_i on the free variables, z_A = 0,
//   N*-operator  r = N* n+  : r_a = -sigma_a (H z)_a for the active variables,
// and H_FF is re-factorised (unrolled Cholesky with masked rows, <= NJ^3/6 fma) at every active-set change instead of
// updating QuadProg++'s J / R factors: with NJ <= 10 that is cheaper than keeping the factors in registers.
// For a positive-definite H the minimiser is unique -- the iterates equal the reference's up to rounding.  H = J'WJ that
// is not positive definite (more than 6 input joints, or a singular pose) is reported per pose (status -1); the
// reference hands such a matrix to a Cholesky factorisation regardless.
#include <hip/hip_runtime.h>
#include <cfloat>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"

namespace
{

// a Cholesky pivot of H below this fraction of trace(H) counts as "not positive definite": a rank-deficient J'WJ leaves
// rounding-level pivots of either sign (same rule in the test oracle)
#define RDYN_IK_PIVOT_FLOOR 1e-10
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))  // lower triangle, i >= j

// Cholesky of the matrix that equals H on the free variables and the identity on the variables in `fixed`, then
// solves for rhs (entries of fixed variables must be 0).  Returns false when a free pivot is not above `floor`.
// c2 accumulates sum 1 / L_ii over the variables NOT in `skip` (QuadProg++'s trace of J = L^-T).
template <int NJ>
__device__ __forceinline__ bool masked_solve(const double (&H)[NJ * (NJ + 1) / 2], unsigned fixed, const double (&rhs)[NJ],
                                             double (&x)[NJ], unsigned skip, double floor, double& c2)
{
  double L[NJ * (NJ + 1) / 2], inv[NJ];
  bool ok = true;
  c2 = 0.0;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
  {
    const bool fj = (fixed >> j) & 1u;
    double d = fj ? 1.0 : H[TRI(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d = fma(-L[TRI(j, k)], L[TRI(j, k)], d);
    ok = ok && (fj || d > floor);
    const double sd = sqrt(d);
    inv[j] = 1.0 / sd;
    L[TRI(j, j)] = sd;
    if (!((skip >> j) & 1u)) c2 += inv[j];
#pragma unroll
    for (int i = j + 1; i < NJ; ++i)
    {
      const bool fi = (fixed >> i) & 1u;
      double v = (fi || fj) ? 0.0 : H[TRI(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-L[TRI(i, k)], L[TRI(j, k)], v);
      L[TRI(i, j)] = v * inv[j];
    }
  }
  double y[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i)
  {
    double v = rhs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v = fma(-L[TRI(i, k)], y[k], v);
    y[i] = v * inv[i];
  }
#pragma unroll
  for (int i = NJ - 1; i >= 0; --i)
  {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < NJ; ++k) v = fma(-L[TRI(k, i)], x[k], v);
    x[i] = v * inv[i];
  }
  return ok;
}

// Eigen::Quaterniond(R) (Eigen 3.3/3.4: trace / largest-diagonal branches), R row-major
__device__ __forceinline__ void quaternion_of(const double (&M)[9], double& qx, double& qy, double& qz, double& qw)
{
  double t = M[0] + M[4] + M[8];
  if (t > 0.0)
  {
    t = sqrt(t + 1.0);
    qw = 0.5 * t;
    t = 0.5 / t;
    qx = (M[7] - M[5]) * t;
    qy = (M[2] - M[6]) * t;
    qz = (M[3] - M[1]) * t;
  }
  else if (M[0] >= M[4] && M[0] >= M[8])  // i = 0
  {
    t = sqrt(M[0] - M[4] - M[8] + 1.0);
    qx = 0.5 * t;
    t = 0.5 / t;
    qw = (M[7] - M[5]) * t;
    qy = (M[3] + M[1]) * t;
    qz = (M[6] + M[2]) * t;
  }
  else if (M[4] >= M[8])  // i = 1
  {
    t = sqrt(M[4] - M[8] - M[0] + 1.0);
    qy = 0.5 * t;
    t = 0.5 / t;
    qw = (M[2] - M[6]) * t;
    qz = (M[7] + M[5]) * t;
    qx = (M[1] + M[3]) * t;
  }
  else  // i = 2
  {
    t = sqrt(M[8] - M[0] - M[4] + 1.0);
    qz = 0.5 * t;
    t = 0.5 / t;
    qw = (M[3] - M[1]) * t;
    qx = (M[2] + M[6]) * t;
    qy = (M[5] + M[7]) * t;
  }
}

// Eigen::AngleAxisd(R).angle() * .axis()  (Eigen 3.3/3.4: quaternion -> angle in [0, pi], axis = sign(w) vec / |vec|)
__device__ __forceinline__ V3 rotation_vector(const double (&M)[9])
{
  double qx, qy, qz, qw;
  quaternion_of(M, qx, qy, qz, qw);
  const double n = sqrt(fma(qx, qx, fma(qy, qy, qz * qz)));
  if (n == 0.0) return mk(0, 0, 0);
  const double k = 2.0 * atan2(n, fabs(qw)) / (qw < 0.0 ? -n : n);
  return mk(qx * k, qy * k, qz * k);
}

// One pose: the iteration from update number it0 on, starting at `start` (the seeds, or the iterate a previous launch
// left in a.sol); writes sol / status / iterations of pose s.
template <int NJ>
__device__ __forceinline__ void ik_pose(const RdynIkArgs& a, ChainPtr c, const int64_t s, const double* __restrict__ start, const int it0)
{
  constexpr int NH = NJ * (NJ + 1) / 2;

  // target frame, column-major 3x4 [R | p] (the record rdyn_transformation writes)
  double Ra[9];  // row-major
  const double* __restrict__ tp = a.T_target + s * a.tt_ss;
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int r = 0; r < 3; ++r) Ra[r * 3 + cc] = tp[(int64_t)(cc * 3 + r) * a.tt_se];
  const V3 pa = mk(tp[9 * a.tt_se], tp[10 * a.tt_se], tp[11 * a.tt_se]);

  // joints that never move in the QP: fixed ones and those outside the input list
  unsigned perm = 0;
  int n_active = 0;
  double sol[NJ];
#pragma unroll
  for (int l = 0; l < NJ; ++l)
  {
    const int idx = c->j[l].in_idx;
    const bool moves = idx >= 0 && c->j[l].type != RDYN_FIXED;
    sol[l] = idx >= 0 ? start[s * a.in_ss + idx * a.in_sj] : 0.0;  // :1403
    if (!moves) perm |= 1u << l;
    if (idx >= 0) ++n_active;
  }
  const double w0 = a.weight[0], w1 = a.weight[1], w2 = a.weight[2], w3 = a.weight[3], w4 = a.weight[4], w5 = a.weight[5];

  int status = 0, it = it0;
  for (;; ++it)
  {
    // ---- frames and screws at sol (computeFrames / computeScrews, primitives_impl.h:863-882)
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3 p = mk(0, 0, 0);
    V3 z[NJ], po[NJ];
#pragma unroll
    for (int f = 0; f < NJ; ++f)
    {
      JointRef J = c->j[f];
      const int type = J.type;
      double Rpc[9];
      V3 t = ld3(J.t);
      if (type == RDYN_REVOLUTE)
      {
        double sn, cs;
        sincos(sol[f], &sn, &cs);
        const double oc = 1.0 - cs;
#pragma unroll
        for (int i = 0; i < 9; ++i) Rpc[i] = fma(sn, J.B[i], fma(oc, J.C[i], J.A[i]));
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 9; ++i) Rpc[i] = J.A[i];
        if (type == RDYN_PRISMATIC) t = axpy(t, ld3(J.up), sol[f]);
      }
      z[f] = rot(R, ld3(J.up));
      p = p + rot(R, t);
      po[f] = p;
      double Rn[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          Rn[r * 3 + cc] = fma(R[r * 3 + 0], Rpc[cc], fma(R[r * 3 + 1], Rpc[3 + cc], R[r * 3 + 2] * Rpc[6 + cc]));
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    if (a.has_tail)
    {
      // a chain served through its reduced companion: the constant frames behind the last input joint
      p = p + rot(R, mk(a.tail_t[0], a.tail_t[1], a.tail_t[2]));
      double Rn[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          Rn[r * 3 + cc] = fma(R[r * 3 + 0], a.tail_R[cc], fma(R[r * 3 + 1], a.tail_R[3 + cc], R[r * 3 + 2] * a.tail_R[6 + cc]));
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    // ---- getFrameDistance(T_target, T_bt): e = [p_a - p_b ; -R_a * (angle * axis)(R_a^T R_b)]   (frame_distance.h:44-49)
    double Rab[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Rab[r * 3 + cc] = fma(Ra[0 + r], R[0 + cc], fma(Ra[3 + r], R[3 + cc], Ra[6 + r] * R[6 + cc]));
    const V3 el = pa - p;
    const V3 rv = rot(Ra, rotation_vector(Rab));
    const V3 ea = mk(-rv.x, -rv.y, -rv.z);
    const V3 wl = mk(w0 * el.x, w1 * el.y, w2 * el.z), wa = mk(w3 * ea.x, w4 * ea.y, w5 * ea.z);
    // computeLocalIk tests |e|, computeWeigthedLocalIk |w o e| (:1409, :1446); weight = 1 in the former
    if (sqrt(dot(wl, wl) + dot(wa, wa)) < a.toll)
    {
      status = 1;
      break;
    }
    if (it >= a.max_iter) break;

    // ---- H = J' W J, g = -J' W e in chain-joint coordinates (:1414-1415, :1451-1452); columns of immovable joints are 0
    V3 jl[NJ], ja[NJ];
#pragma unroll
    for (int l = 0; l < NJ; ++l)
    {
      const int type = c->j[l].type;
      const bool moves = !((perm >> l) & 1u);
      jl[l] = mk(0, 0, 0);
      ja[l] = mk(0, 0, 0);
      if (moves && type == RDYN_REVOLUTE)
      {
        jl[l] = cross(z[l], p - po[l]);  // spatialTranslation(S_l, p_tool - p_l), :944
        ja[l] = z[l];
      }
      else if (moves && type == RDYN_PRISMATIC)
        jl[l] = z[l];
    }
    double H[NH], g[NJ], lo[NJ], hi[NJ], x[NJ];
    double c1 = 0.0;
#pragma unroll
    for (int i = 0; i < NJ; ++i)
    {
      const V3 wjl = mk(w0 * jl[i].x, w1 * jl[i].y, w2 * jl[i].z), wja = mk(w3 * ja[i].x, w4 * ja[i].y, w5 * ja[i].z);
      g[i] = -(dot(wjl, el) + dot(wja, ea));
#pragma unroll
      for (int j = 0; j <= i; ++j) H[TRI(i, j)] = dot(wjl, jl[j]) + dot(wja, ja[j]);
      if (!((perm >> i) & 1u))
      {
        H[TRI(i, i)] += a.damping * a.damping;  // 0 unless the caller asked for a damped step (rdyn_local_ik_damped)
        c1 += H[TRI(i, i)];
      }
      lo[i] = a.q_min[i] - sol[i];  // dq_i >= lo_i  <=>  ci0 = sol - q_min (:1417)
      hi[i] = a.q_max[i] - sol[i];  // dq_i <= hi_i  <=>  ci0 = q_max - sol (:1418)
    }

    // ---- Goldfarb-Idnani, bound-constrained
    double rhs[NJ], c2, c2_unused;
#pragma unroll
    for (int i = 0; i < NJ; ++i) rhs[i] = ((perm >> i) & 1u) ? 0.0 : -g[i];
    if (!masked_solve<NJ>(H, perm, rhs, x, perm, RDYN_IK_PIVOT_FLOOR * c1, c2))
    {
      status = -1;  // H not positive definite
      break;
    }
    const double thr = (2.0 * n_active) * DBL_EPSILON * c1 * c2 * 100.0;
    unsigned actL = 0, actU = 0;
    double u[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) u[i] = 0.0;
    int qp = 0;  // 0 solved
    for (int guard = 0;; ++guard)
    {
      if (guard >= 50 * (2 * NJ + 1))
      {
        qp = -3;
        break;
      }
      // step 1: the most violated bound
      double psi = 0.0, ss = 0.0;
      int ip = -1;  // variable; upper bounds are ip + NJ
#pragma unroll
      for (int i = 0; i < NJ; ++i)
      {
        const bool cand = !(((perm | actL) >> i) & 1u);
        const double sl = cand ? x[i] - lo[i] : 0.0;
        psi += fmin(0.0, sl);
        if (sl < ss) { ss = sl; ip = i; }
      }
#pragma unroll
      for (int i = 0; i < NJ; ++i)
      {
        const bool cand = !(((perm | actU) >> i) & 1u);
        const double su = cand ? hi[i] - x[i] : 0.0;
        psi += fmin(0.0, su);
        if (su < ss) { ss = su; ip = i + NJ; }
      }
      if (fabs(psi) <= thr || ip < 0) break;
      const bool upper = ip >= NJ;
      const int iv = upper ? ip - NJ : ip;
      const double sg = upper ? -1.0 : 1.0;
      if (((actL | actU) >> iv) & 1u)
      {
        qp = -2;  // both bounds of one variable: q_min > q_max
        break;
      }
      double uplus = 0.0, sip = ss;
      for (int inner = 0;; ++inner)
      {
        if (inner >= 4 * (2 * NJ + 1))
        {
          qp = -3;
          break;
        }
        // step 2a: z = H n+ on the free variables, r = N* n+ on the active ones
        const unsigned fixed = perm | actL | actU;
        double zz[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i) rhs[i] = (i == iv) ? sg : 0.0;
        masked_solve<NJ>(H, fixed, rhs, zz, fixed, 0.0, c2_unused);
        double t1 = INFINITY, zi = 0.0;
        int l = -1;
        double r[NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i)
        {
          double hz = 0.0;
#pragma unroll
          for (int j = 0; j < NJ; ++j) hz = fma(H[i >= j ? TRI(i, j) : TRI(j, i)], zz[j], hz);
          const bool lowA = (actL >> i) & 1u, upA = (actU >> i) & 1u;
          r[i] = lowA ? -hz : (upA ? hz : 0.0);
          if ((lowA || upA) && r[i] > 0.0 && u[i] / r[i] < t1)
          {
            t1 = u[i] / r[i];
            l = i;
          }
          if (i == iv) zi = zz[i];
        }
        // step 2b: z'n+ = sg * z_iv > 0 for a positive-definite H_FF
        const double t2 = -sip / (sg * zi);
        const double t = fmin(t1, t2);
        // step 2c
#pragma unroll
        for (int i = 0; i < NJ; ++i)
        {
          x[i] = fma(t, zz[i], x[i]);
          u[i] = fma(-t, r[i], u[i]);
        }
        uplus += t;
        if (t == t2)
        {
          if (upper) actU |= 1u << iv;
          else actL |= 1u << iv;
#pragma unroll
          for (int i = 0; i < NJ; ++i)
            if (i == iv) u[i] = uplus;
          break;
        }
        actL &= ~(1u << l);
        actU &= ~(1u << l);
#pragma unroll
        for (int i = 0; i < NJ; ++i)
        {
          if (i == l) u[i] = 0.0;
          if (i == iv) sip = upper ? hi[i] - x[i] : x[i] - lo[i];
        }
      }
      if (qp != 0) break;
    }
    if (qp != 0)
    {
      status = qp;
      break;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) sol[i] += x[i];  // :1428
  }

#pragma unroll
  for (int l = 0; l < NJ; ++l)
  {
    const int idx = c->j[l].in_idx;
    if (idx >= 0) a.sol[s * a.in_ss + idx * a.in_sj] = sol[l];
  }
  if (a.status) a.status[s] = status;
  if (a.iterations) a.iterations[s] = it;
}

template <int NJ>
__global__ __launch_bounds__(64) void k_local_ik(const RdynIkArgs a)
{
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= a.n_samples) return;
  ik_pose<NJ>(a, as_const(a.chain), s, a.seed, 0);
}

// Second stage: poses need very different numbers of updates (most converge within a few, some never do), and a wave
// runs as long as its slowest lane.  After a first launch capped at a.it_stage updates, this kernel scans the statuses
// in chunks of RDYN_IK_CHUNK poses, gathers the poses that are still running (status 0 with exactly it_stage updates)
// into a dense list in LDS and continues ONLY those, 64 per pass, from the iterate the first stage left in a.sol.
// Per pose the arithmetic is the same sequence of updates as in a single launch.
#define RDYN_IK_CHUNK 1024
template <int NJ>
__global__ __launch_bounds__(64) void k_local_ik_resume(const RdynIkArgs a)
{
  __shared__ int list[RDYN_IK_CHUNK];
  __shared__ int count;
  ChainPtr c = as_const(a.chain);
  const int64_t n_chunks = (a.n_samples + RDYN_IK_CHUNK - 1) / RDYN_IK_CHUNK;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x)
  {
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    const int64_t base = chunk * RDYN_IK_CHUNK;
    for (int i = threadIdx.x; i < RDYN_IK_CHUNK; i += 64)
    {
      const int64_t s = base + i;
      if (s < a.n_samples && a.status[s] == 0 && a.iterations[s] == a.it_stage) list[atomicAdd(&count, 1)] = i;
    }
    __syncthreads();
    const int cnt = count;
    for (int g0 = 0; g0 < cnt; g0 += 64)
      if (g0 + (int)threadIdx.x < cnt) ik_pose<NJ>(a, c, base + list[g0 + threadIdx.x], a.sol, a.it_stage);
    __syncthreads();
  }
}

// getFrameDistance (kind 0), getFrameDistanceQuat (1), getFrameDistanceQuatJac (2): frame_distance.h:44-49, 73-86, 112-126.
// One thread per pair of frames (column-major 3x4 [R | p] records); the Jacobian is 6 x 6 column-major.
__global__ __launch_bounds__(256) void k_frame_distance(const RdynFrameDistanceArgs a)
{
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.n) return;
  double Ra[9], Rb[9];
  const double* __restrict__ ta = a.T_wa + s * a.t_ss;
  const double* __restrict__ tb = a.T_wb + s * a.t_ss;
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int r = 0; r < 3; ++r)
    {
      Ra[r * 3 + cc] = ta[(int64_t)(cc * 3 + r) * a.t_se];
      Rb[r * 3 + cc] = tb[(int64_t)(cc * 3 + r) * a.t_se];
    }
  const V3 pa = mk(ta[9 * a.t_se], ta[10 * a.t_se], ta[11 * a.t_se]), pb = mk(tb[9 * a.t_se], tb[10 * a.t_se], tb[11 * a.t_se]);
  double Rab[9];  // R_wa^-1 R_wb
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) Rab[r * 3 + cc] = fma(Ra[0 + r], Rb[0 + cc], fma(Ra[3 + r], Rb[3 + cc], Ra[6 + r] * Rb[6 + cc]));
  V3 lin = pa - pb, ang;
  double qx = 0.0, qy = 0.0, qz = 0.0, qw = 1.0;
  if (a.kind == 0)
  {
    const V3 rv = rot(Ra, rotation_vector(Rab));  // -R_wa (angle * axis), :48
    ang = mk(-rv.x, -rv.y, -rv.z);
  }
  else
  {
    quaternion_of(Rab, qx, qy, qz, qw);
    if (qw < 0.0)  // :77-83, :118-122
    {
      qx = -qx; qy = -qy; qz = -qz; qw = -qw;
    }
    const V3 rv = rot(Ra, mk(qx, qy, qz));        // -2 R_wa imag(q_ab), :84 / :124
    ang = mk(-2.0 * rv.x, -2.0 * rv.y, -2.0 * rv.z);
    if (a.kind == 2) lin = pb - pa;               // the Jacobian variant measures the translation the other way round (:115)
  }
  double* __restrict__ o = a.distance + s * a.d_ss;
  o[0] = lin.x; o[a.d_se] = lin.y; o[2 * a.d_se] = lin.z;
  o[3 * a.d_se] = ang.x; o[4 * a.d_se] = ang.y; o[5 * a.d_se] = ang.z;
  if (a.kind == 2 && a.jacobian)
  {
    // J = [I 0; 0 R_wa (w I - skew(vec)) R_wa^T], :125
    const double K[9] = {qw, qz, -qy, -qz, qw, qx, qy, -qx, qw};  // w I - skew(v), row-major
    double RK[9], B[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) RK[r * 3 + cc] = fma(Ra[r * 3], K[cc], fma(Ra[r * 3 + 1], K[3 + cc], Ra[r * 3 + 2] * K[6 + cc]));
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) B[r * 3 + cc] = fma(RK[r * 3], Ra[cc * 3], fma(RK[r * 3 + 1], Ra[cc * 3 + 1], RK[r * 3 + 2] * Ra[cc * 3 + 2]));
    double* __restrict__ jo = a.jacobian + s * a.j_ss;
#pragma unroll
    for (int cc = 0; cc < 6; ++cc)
#pragma unroll
      for (int r = 0; r < 6; ++r)
      {
        double v = (r == cc) ? 1.0 : 0.0;
        if (r >= 3 && cc >= 3) v = B[(r - 3) * 3 + (cc - 3)];
        jo[(int64_t)(cc * 6 + r) * a.j_se] = v;
      }
  }
}

template <int NJ>
hipError_t launch_ik_nj(const RdynIkArgs& a, hipStream_t st)
{
  const unsigned grid = (unsigned)((a.n_samples + 63) / 64);
  if (a.it_stage > 0)
  {
    const int64_t n_chunks = (a.n_samples + RDYN_IK_CHUNK - 1) / RDYN_IK_CHUNK;
    hipLaunchKernelGGL((k_local_ik_resume<NJ>), dim3((unsigned)(n_chunks < 65536 ? n_chunks : 65536)), dim3(64), 0, st, a);
  }
  else
    hipLaunchKernelGGL((k_local_ik<NJ>), dim3(grid), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace

hipError_t rdyn_launch_frame_distance(const RdynFrameDistanceArgs& a, hipStream_t st)
{
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_frame_distance, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t rdyn_launch_local_ik(int n_joints, const RdynIkArgs& a, hipStream_t st)
{
  switch (n_joints)
  {
  case 1: return launch_ik_nj<1>(a, st);
  case 2: return launch_ik_nj<2>(a, st);
  case 3: return launch_ik_nj<3>(a, st);
  case 4: return launch_ik_nj<4>(a, st);
  case 5: return launch_ik_nj<5>(a, st);
  case 6: return launch_ik_nj<6>(a, st);
  case 7: return launch_ik_nj<7>(a, st);
  case 8: return launch_ik_nj<8>(a, st);
  case 9: return launch_ik_nj<9>(a, st);
  case 10: return launch_ik_nj<10>(a, st);
  default: return hipErrorInvalidValue;
  }
}
