/*
 * ik_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rosdyn_oracle.c; PARITY UNPINNED at the reference level).
 *
 * CPU restatement of the local inverse kinematics of rosdyn_core (paths under /root/reference/rosdyn_core/include/rosdyn_core/):
 *   getFrameDistance               frame_distance.h:44-49
 *   Chain::computeLocalIk          internal/primitives_impl.h:1398-1433
 *   Chain::computeWeigthedLocalIk  internal/primitives_impl.h:1436-1468
 * built on the public per-sample entry points of rosdyn_oracle.c (orc_fk, orc_jacobian).
 *
 * Third-party arithmetic that is NOT under /root/reference and is restated here from its published algorithm
 * (no version is pinned by the reference: rosdyn.rosinstall lists the upstream repositories without tags):
 *   - Eigen::solve_quadprog (eigen_matrix_utils' eiquadprog.hpp, primitives.h:53; a port of QuadProg++): the
 *     Goldfarb-Idnani dual active-set method, "A numerically stable dual method for solving strictly convex quadratic
 *     programs", Math. Programming 27 (1983).  Problem form: min 1/2 x'Gx + g0'x  s.t.  CI'x + ci0 >= 0 (CE is n x 0 on
 *     this path, primitives_impl.h:787-788).  The operators of the paper (H = reduced inverse Hessian, N* = pseudo-inverse
 *     of the active normals in the G^-1 metric) are evaluated here from their DEFINITIONS by dense solves at every step
 *     instead of through the J / R factor updates of QuadProg++; the iterates are the same in exact arithmetic.
 *     For a positive-definite G the minimiser is unique, which is what the parity tests compare.
 *     A G that is not positive definite (n > 6, or a singular configuration: G = J'J) makes the Cholesky factorisation
 *     of QuadProg++ meaningless; here it is reported as status -1 (pivot <= 1e-10 trace(G)).
 *   - Eigen::AngleAxisd(Matrix3d) (Eigen 3.3 / 3.4 Geometry): rotation matrix -> quaternion (trace / largest-diagonal
 *     branches), quaternion -> angle in [0, pi] = 2 atan2(|vec|, |w|), axis = sign(w) vec / |vec|.
 * The reference bounds the loop by wall-clock time (ros::Duration max_time, default 5 ms); here it is an iteration cap.
 */
#include <float.h>
#include <math.h>
#include <string.h>

#define IK_MAX_N 16
#define IK_MAX_M (2 * IK_MAX_N)
/* A Cholesky pivot of G below IK_PIVOT_FLOOR * trace(G) is treated as "not positive definite": a rank-deficient J'WJ
 * leaves pivots of either sign at rounding level, and QuadProg++'s own test (pivot <= 0) would then depend on noise.
 * The HIP kernel applies the same rule (rdyn_ik.hip). */
#define IK_PIVOT_FLOOR 1e-10

typedef struct orc_chain orc_chain;
void orc_fk(const orc_chain* c, const double* q, double* T_all);
void orc_jacobian(const orc_chain* c, const double* q, double* J);

/* ---------------------------------------------------------------- Eigen::AngleAxisd(R) -> angle * axis */
static void angle_axis_vector(const double R[3][3], double out[3])
{
  double q[4]; /* x, y, z, w */
  double t = R[0][0] + R[1][1] + R[2][2];
  if (t > 0.0)
  {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  }
  else
  {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n != 0.0)
  {
    double angle = 2.0 * atan2(n, fabs(q[3]));
    if (q[3] < 0.0) n = -n;
    for (int i = 0; i < 3; i++) out[i] = angle * (q[i] / n);
  }
  else
    out[0] = out[1] = out[2] = 0.0; /* angle 0 (axis (1,0,0)) */
}

/* getFrameDistance(T_wa, T_wb, distance), frame_distance.h:44-49; T = row-major 3x4 [R|p] */
void orc_frame_distance(const double* T_wa, const double* T_wb, double* distance)
{
  double Rab[3][3], aa[3];
  for (int i = 0; i < 3; i++) distance[i] = T_wa[i * 4 + 3] - T_wb[i * 4 + 3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      double acc = 0; /* R_wa^-1 * R_wb */
      for (int k = 0; k < 3; k++) acc += T_wa[k * 4 + i] * T_wb[k * 4 + j];
      Rab[i][j] = acc;
    }
  angle_axis_vector(Rab, aa);
  for (int i = 0; i < 3; i++)
  {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += T_wa[i * 4 + k] * aa[k];
    distance[3 + i] = -acc;
  }
}

/* getFrameDistanceQuat (jacobian == NULL and jac_variant == 0, frame_distance.h:73-86) and getFrameDistanceQuatJac
 * (jac_variant != 0, frame_distance.h:112-126: the translation part is T_wb - T_wa there; jacobian = 6 x 6 row-major, may
 * be NULL).  T = row-major 3x4 [R|p]. */
void orc_frame_distance_quat(const double* T_wa, const double* T_wb, const double* jac_variant, double* distance, double* jacobian)
{
  double Rab[3][3], q[4];
  const int jv = jac_variant && jac_variant[0] != 0.0;
  for (int i = 0; i < 3; i++) distance[i] = jv ? T_wb[i * 4 + 3] - T_wa[i * 4 + 3] : T_wa[i * 4 + 3] - T_wb[i * 4 + 3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += T_wa[k * 4 + i] * T_wb[k * 4 + j];
      Rab[i][j] = acc;
    }
  { /* Eigen::Quaterniond(Matrix3d), as in angle_axis_vector */
    double t = Rab[0][0] + Rab[1][1] + Rab[2][2];
    if (t > 0.0)
    {
      t = sqrt(t + 1.0);
      q[3] = 0.5 * t;
      t = 0.5 / t;
      q[0] = (Rab[2][1] - Rab[1][2]) * t;
      q[1] = (Rab[0][2] - Rab[2][0]) * t;
      q[2] = (Rab[1][0] - Rab[0][1]) * t;
    }
    else
    {
      int i = 0;
      if (Rab[1][1] > Rab[0][0]) i = 1;
      if (Rab[2][2] > Rab[i][i]) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3;
      t = sqrt(Rab[i][i] - Rab[j][j] - Rab[k][k] + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (Rab[k][j] - Rab[j][k]) * t;
      q[j] = (Rab[j][i] + Rab[i][j]) * t;
      q[k] = (Rab[k][i] + Rab[i][k]) * t;
    }
  }
  if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];                /* :77-83 */
  for (int i = 0; i < 3; i++)
  {
    double acc = 0;
    for (int k = 0; k < 3; k++) acc += T_wa[i * 4 + k] * q[k];
    distance[3 + i] = -2.0 * acc;                                        /* :84 */
  }
  if (jacobian)
  {
    const double K[3][3] = {{q[3], q[2], -q[1]}, {-q[2], q[3], q[0]}, {q[1], -q[0], q[3]}}; /* w I - skew(vec) */
    for (int i = 0; i < 36; i++) jacobian[i] = 0.0;
    for (int i = 0; i < 6; i++) jacobian[i * 6 + i] = 1.0;               /* setIdentity, :114 */
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        double acc = 0;                                                  /* R_wa K R_wa^-1, :125 */
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) acc += T_wa[i * 4 + a] * K[a][b] * T_wa[j * 4 + b];
        jacobian[(3 + i) * 6 + 3 + j] = acc;
      }
  }
}

/* ---------------------------------------------------------------- small dense helpers (row-major, leading dim IK_MAX_N) */
static int cholesky(int n, const double A[IK_MAX_N][IK_MAX_N], double L[IK_MAX_N][IK_MAX_N], double floor)
{
  memset(L, 0, sizeof(double) * IK_MAX_N * IK_MAX_N);
  for (int j = 0; j < n; j++)
  {
    double d = A[j][j];
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    if (!(d > floor)) return 0;
    L[j][j] = sqrt(d);
    for (int i = j + 1; i < n; i++)
    {
      double v = A[i][j];
      for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
      L[i][j] = v / L[j][j];
    }
  }
  return 1;
}
static void chol_solve(int n, const double L[IK_MAX_N][IK_MAX_N], const double* b, double* x)
{
  double y[IK_MAX_N];
  for (int i = 0; i < n; i++)
  {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
    y[i] = v / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--)
  {
    double v = y[i];
    for (int k = i + 1; k < n; k++) v -= L[k][i] * x[k];
    x[i] = v / L[i][i];
  }
}

/* Goldfarb-Idnani.  G n x n (row-major, ld n), g0 n, CI n x m (column i = normal of constraint i, row-major ld m), ci0 m.
 * returns 0 ok, -1 G not positive definite, -2 infeasible, -3 iteration guard. */
int orc_solve_quadprog(int n, int m, const double* G_in, const double* g0, const double* CI, const double* ci0, double* x)
{
  double G[IK_MAX_N][IK_MAX_N], L[IK_MAX_N][IK_MAX_N];
  if (n > IK_MAX_N || m > IK_MAX_M) return -3;
  memset(G, 0, sizeof G);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) G[i][j] = G_in[i * n + j];
  double c1 = 0;
  for (int i = 0; i < n; i++) c1 += G[i][i];
  if (!cholesky(n, G, L, IK_PIVOT_FLOOR * c1)) return -1;
  double c2 = 0;
  for (int i = 0; i < n; i++) c2 += 1.0 / L[i][i]; /* trace of J = L^-T (QuadProg++'s condition estimate) */
  /* step 0: unconstrained minimiser */
  double mg[IK_MAX_N];
  for (int i = 0; i < n; i++) mg[i] = -g0[i];
  chol_solve(n, L, mg, x);
  int A[IK_MAX_M], q = 0;  /* active set */
  double u[IK_MAX_M];
  for (int guard = 0; guard < 50 * (m + 1); guard++)
  {
    /* step 1: most violated constraint */
    double s[IK_MAX_M], psi = 0, ss = 0;
    int ip = -1;
    for (int i = 0; i < m; i++)
    {
      int active = 0;
      for (int k = 0; k < q; k++) active |= (A[k] == i);
      double v = ci0[i];
      for (int k = 0; k < n; k++) v += CI[k * m + i] * x[k];
      s[i] = active ? 0.0 : v;
      psi += fmin(0.0, s[i]);
    }
    if (fabs(psi) <= m * DBL_EPSILON * c1 * c2 * 100.0) return 0;
    for (int i = 0; i < m; i++)
      if (s[i] < ss) { ss = s[i]; ip = i; }
    if (ip < 0) return 0;
    double np[IK_MAX_N], uplus = 0, sip = s[ip];
    for (int k = 0; k < n; k++) np[k] = CI[k * m + ip];
    for (int inner = 0; inner < 4 * (m + 1); inner++)
    {
      /* step 2a: z = H np, r = N* np with N = normals of the active set */
      double Ginp[IK_MAX_N], GiN[IK_MAX_M][IK_MAX_N], z[IK_MAX_N], r[IK_MAX_M];
      chol_solve(n, L, np, Ginp);
      for (int k = 0; k < n; k++) z[k] = Ginp[k];
      if (q > 0)
      {
        double S[IK_MAX_N][IK_MAX_N], LS[IK_MAX_N][IK_MAX_N], rhs[IK_MAX_N], col[IK_MAX_N];
        memset(S, 0, sizeof S);
        for (int a = 0; a < q; a++)
        {
          for (int k = 0; k < n; k++) col[k] = CI[k * m + A[a]];
          chol_solve(n, L, col, GiN[a]);
        }
        for (int a = 0; a < q; a++)
        {
          rhs[a] = 0;
          for (int k = 0; k < n; k++) rhs[a] += CI[k * m + A[a]] * Ginp[k];
          for (int b = 0; b < q; b++)
          {
            double v = 0;
            for (int k = 0; k < n; k++) v += CI[k * m + A[a]] * GiN[b][k];
            S[a][b] = v;
          }
        }
        if (q > IK_MAX_N || !cholesky(q, S, LS, 0.0)) return -2; /* dependent active normals */
        chol_solve(q, LS, rhs, r);
        for (int a = 0; a < q; a++)
          for (int k = 0; k < n; k++) z[k] -= GiN[a][k] * r[a];
      }
      /* step 2b: step lengths */
      double t1 = INFINITY, t2 = INFINITY, zz = 0, znp = 0;
      int l = -1;
      for (int a = 0; a < q; a++)
        if (r[a] > 0.0 && u[a] / r[a] < t1) { t1 = u[a] / r[a]; l = a; }
      for (int k = 0; k < n; k++) { zz += z[k] * z[k]; znp += z[k] * np[k]; }
      if (fabs(zz) > DBL_EPSILON) t2 = -sip / znp;
      double t = fmin(t1, t2);
      if (t >= INFINITY) return -2;
      /* step 2c */
      if (t2 >= INFINITY)
      {
        for (int a = 0; a < q; a++) u[a] -= t * r[a];
        uplus += t;
        for (int a = l; a + 1 < q; a++) { A[a] = A[a + 1]; u[a] = u[a + 1]; }
        q--;
        continue;
      }
      for (int k = 0; k < n; k++) x[k] += t * z[k];
      for (int a = 0; a < q; a++) u[a] -= t * r[a];
      uplus += t;
      if (t == t2)
      {
        A[q] = ip;
        u[q] = uplus;
        q++;
        break;
      }
      for (int a = l; a + 1 < q; a++) { A[a] = A[a + 1]; u[a] = u[a + 1]; }
      q--;
      sip = ci0[ip];
      for (int k = 0; k < n; k++) sip += np[k] * x[k];
    }
  }
  return -3;
}

/* computeLocalIk (weight == NULL, primitives_impl.h:1398-1433) / computeWeigthedLocalIk (1436-1468).
 * T_target: row-major 3x4; q_min / q_max / seed / sol: n_active; iterations (out): QP updates performed.
 * returns 1 converged (the reference's `true`), 0 not converged within max_iter updates (`false` after max_time),
 * -1 / -2 / -3 the QP failed (see orc_solve_quadprog) -- sol then holds the last iterate. */
int orc_local_ik(const orc_chain* c, int n_active, int n_links, const double* T_target, const double* seed, const double* weight,
                 const double* q_min, const double* q_max, double toll, double damping, int max_iter, double* sol, int* iterations)
{
  /* damping: not in the reference -- the product's Levenberg option (rdyn_local_ik_damped), damping^2 on the diagonal of H */
  const int n = n_active, m = 2 * n_active;
  double T_all[IK_MAX_N * 12 + 12], J[6 * IK_MAX_N], e[6], H[IK_MAX_N * IK_MAX_N], f[IK_MAX_N], CI[IK_MAX_N * IK_MAX_M], ci0[IK_MAX_M],
      dq[IK_MAX_N];
  memcpy(sol, seed, sizeof(double) * n);                                  /* :1403 */
  memset(CI, 0, sizeof CI);                                               /* m_CI = [I, -I], :792-793 */
  for (int i = 0; i < n; i++) { CI[i * m + i] = 1.0; CI[i * m + n + i] = -1.0; }
  for (int it = 0;; it++)
  {
    *iterations = it;
    orc_fk(c, sol, T_all);
    orc_frame_distance(T_target, T_all + 12 * (n_links - 1), e);         /* :1408 */
    double nrm = 0;
    for (int i = 0; i < 6; i++) { double v = weight ? weight[i] * e[i] : e[i]; nrm += v * v; }
    if (sqrt(nrm) < toll) return 1;                                        /* :1409-1412, :1446-1449 */
    if (it >= max_iter) return 0;
    orc_jacobian(c, sol, J);                                               /* column-major 6 x n */
    for (int a = 0; a < n; a++)
    {
      f[a] = 0;
      for (int i = 0; i < 6; i++) f[a] -= J[a * 6 + i] * (weight ? weight[i] : 1.0) * e[i];          /* :1415, :1452 */
      for (int b = 0; b < n; b++)
      {
        double v = 0;
        for (int i = 0; i < 6; i++) v += J[a * 6 + i] * (weight ? weight[i] : 1.0) * J[b * 6 + i];   /* :1414, :1451 */
        H[a * n + b] = v;
      }
      H[a * n + a] += damping * damping;
    }
    for (int i = 0; i < n; i++) { ci0[i] = sol[i] - q_min[i]; ci0[n + i] = q_max[i] - sol[i]; }        /* :1417-1418 */
    int st = orc_solve_quadprog(n, m, H, f, CI, ci0, dq);                  /* :1421-1427 */
    if (st != 0) return st;
    for (int i = 0; i < n; i++) sol[i] += dq[i];                           /* :1428 */
  }
}
