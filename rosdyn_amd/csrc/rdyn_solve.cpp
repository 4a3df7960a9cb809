// rdyn_solve.cpp -- the small dense problems behind the identification step (host, P <= a few hundred unknowns):
//   rdyn_solve_normal_equations   minimum-norm least-squares solution of G x = c, G = A'A symmetric positive SEMI-definite
//                                 (the stacked regressor is structurally rank deficient: unobservable base-link parameters,
//                                 fixed tail links), by a cyclic Jacobi eigen-decomposition truncated at rtol * lambda_max;
//   rdyn_gram_r_factor            rank-revealing R factor of A from its Gram (pivoted Cholesky): R'R = G[perm][:, perm];
//   rdyn_solve_r_factor           minimum-norm solution of min |R x - d| for an upper-triangular / trapezoidal R (the output
//                                 of the TSQR path, which never squares the condition number), by a one-sided Jacobi SVD.
// No reference counterpart inside rosdyn_core: the identification lived in the external rosdyn_identification (README.md:15).
// Plain C++17, no dependencies; everything is O(P^3) with P <= 111 on this path (milliseconds).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "rdyn_chain.hpp"

namespace
{

// cyclic Jacobi on a symmetric n x n matrix (column-major, both triangles used); V gets the eigenvectors in its columns.
// Converges quadratically; 30 sweeps are far more than fp64 needs at n ~ 100.
void jacobi_eigh(std::vector<double>& A, int n, std::vector<double>& V, std::vector<double>& w)
{
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  auto a = [&](int i, int j) -> double& { return A[(size_t)j * n + i]; };
  auto v = [&](int i, int j) -> double& { return V[(size_t)j * n + i]; };
  for (int sweep = 0; sweep < 30; ++sweep)
  {
    double off = 0.0, diag = 0.0;
    for (int j = 0; j < n; ++j)
    {
      diag += a(j, j) * a(j, j);
      for (int i = 0; i < j; ++i) off += a(i, j) * a(i, j);
    }
    if (off <= 1e-34 * (diag + off)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        const double apq = a(p, q);
        if (apq == 0.0) continue;
        const double theta = (a(q, q) - a(p, p)) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k)  // columns p, q
        {
          const double akp = a(k, p), akq = a(k, q);
          a(k, p) = cs * akp - sn * akq;
          a(k, q) = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k)  // rows p, q
        {
          const double apk = a(p, k), aqk = a(q, k);
          a(p, k) = cs * apk - sn * aqk;
          a(q, k) = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k)
        {
          const double vkp = v(k, p), vkq = v(k, q);
          v(k, p) = cs * vkp - sn * vkq;
          v(k, q) = sn * vkp + cs * vkq;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = a(i, i);
}

}  // namespace

extern "C"
{

int rdyn_solve_normal_equations(const double* G, const double* c, int n, double rtol, double* x, int* rank)
{
  if (!G || !c || !x || n < 1 || !(rtol >= 0.0))
  {
    rdyn_set_error("rdyn_solve_normal_equations: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  std::vector<double> A((size_t)n * n), V, w;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) A[(size_t)j * n + i] = 0.5 * (G[(size_t)j * n + i] + G[(size_t)i * n + j]);  // symmetrise
  jacobi_eigh(A, n, V, w);
  double wmax = 0.0;
  for (int i = 0; i < n; ++i) wmax = std::max(wmax, w[i]);
  int r = 0;
  std::fill(x, x + n, 0.0);
  for (int k = 0; k < n; ++k)
  {
    if (!(w[k] > rtol * wmax) || !(w[k] > 0.0)) continue;
    ++r;
    double proj = 0.0;
    for (int i = 0; i < n; ++i) proj += V[(size_t)k * n + i] * c[i];
    proj /= w[k];
    for (int i = 0; i < n; ++i) x[i] += V[(size_t)k * n + i] * proj;
  }
  if (rank) *rank = r;
  return RDYN_OK;
}

int rdyn_gram_r_factor(const double* G, int n, double rtol, double* R, int32_t* perm, int* rank)
{
  if (!G || !R || !perm || n < 1 || !(rtol >= 0.0))
  {
    rdyn_set_error("rdyn_gram_r_factor: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  // pivoted Cholesky, row by row; R is n x n column-major, rows >= rank are zero
  std::vector<double> d(n);
  std::memset(R, 0, sizeof(double) * (size_t)n * n);
  auto g = [&](int i, int j) { return 0.5 * (G[(size_t)j * n + i] + G[(size_t)i * n + j]); };
  auto r_ = [&](int i, int j) -> double& { return R[(size_t)j * n + i]; };
  for (int i = 0; i < n; ++i)
  {
    perm[i] = i;
    d[i] = g(i, i);
  }
  double dmax = 0.0;
  for (int i = 0; i < n; ++i) dmax = std::max(dmax, d[i]);
  int rk = 0;
  for (int k = 0; k < n; ++k)
  {
    int piv = k;
    for (int j = k + 1; j < n; ++j)
      if (d[j] > d[piv]) piv = j;
    if (!(d[piv] > rtol * dmax) || !(d[piv] > 0.0)) break;
    if (piv != k)
    {
      std::swap(perm[k], perm[piv]);
      std::swap(d[k], d[piv]);
      for (int i = 0; i < n; ++i) std::swap(r_(i, k), r_(i, piv));
    }
    const double rkk = std::sqrt(d[k]);
    r_(k, k) = rkk;
    for (int j = k + 1; j < n; ++j)
    {
      double s = g(perm[k], perm[j]);
      for (int i = 0; i < k; ++i) s -= r_(i, k) * r_(i, j);
      const double v = s / rkk;
      r_(k, j) = v;
      d[j] -= v * v;
    }
    ++rk;
  }
  if (rank) *rank = rk;
  return RDYN_OK;
}

int rdyn_solve_r_factor(const double* R, int64_t ldr, int rows, int n, const double* d, double rtol, double* x, int* rank)
{
  if (!R || !d || !x || n < 1 || rows < 1 || ldr < rows || !(rtol >= 0.0))
  {
    rdyn_set_error("rdyn_solve_r_factor: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  // one-sided Jacobi (Hestenes) SVD of R (rows x n): rotate column pairs of U = R V until they are orthogonal;
  // then R = (U D^-1) D V' with D = column norms.  x = V D^-1 (U D^-1)' d restricted to D > rtol * Dmax.
  std::vector<double> U((size_t)rows * n), V((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j)
  {
    for (int i = 0; i < rows; ++i) U[(size_t)j * rows + i] = R[(size_t)j * ldr + i];
    V[(size_t)j * n + j] = 1.0;
  }
  for (int sweep = 0; sweep < 40; ++sweep)
  {
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        double app = 0, aqq = 0, apq = 0;
        const double* up = &U[(size_t)p * rows];
        const double* uq = &U[(size_t)q * rows];
        for (int i = 0; i < rows; ++i)
        {
          app += up[i] * up[i];
          aqq += uq[i] * uq[i];
          apq += up[i] * uq[i];
        }
        if (std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq) || apq == 0.0) continue;
        rotated = true;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        double* wp = &U[(size_t)p * rows];
        double* wq = &U[(size_t)q * rows];
        for (int i = 0; i < rows; ++i)
        {
          const double a = wp[i], b = wq[i];
          wp[i] = cs * a - sn * b;
          wq[i] = sn * a + cs * b;
        }
        double* vp = &V[(size_t)p * n];
        double* vq = &V[(size_t)q * n];
        for (int i = 0; i < n; ++i)
        {
          const double a = vp[i], b = vq[i];
          vp[i] = cs * a - sn * b;
          vq[i] = sn * a + cs * b;
        }
      }
    if (!rotated) break;
  }
  std::vector<double> sv(n);
  double smax = 0.0;
  for (int j = 0; j < n; ++j)
  {
    double s = 0;
    for (int i = 0; i < rows; ++i) s += U[(size_t)j * rows + i] * U[(size_t)j * rows + i];
    sv[j] = std::sqrt(s);
    smax = std::max(smax, sv[j]);
  }
  std::fill(x, x + n, 0.0);
  int rk = 0;
  for (int j = 0; j < n; ++j)
  {
    if (!(sv[j] > rtol * smax) || !(sv[j] > 0.0)) continue;
    ++rk;
    double proj = 0;
    for (int i = 0; i < rows; ++i) proj += U[(size_t)j * rows + i] * d[i];
    proj /= sv[j] * sv[j];
    for (int i = 0; i < n; ++i) x[i] += V[(size_t)j * n + i] * proj;
  }
  if (rank) *rank = rk;
  return RDYN_OK;
}

int rdyn_tsqr_combine_host(const double* R_stack, int n_factors, int n, double* R_out)
{
  if (!R_stack || !R_out || n_factors < 1 || n < 1)
  {
    rdyn_set_error("rdyn_tsqr_combine_host: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  // Householder QR of the stacked factors (n_factors * n rows, n columns), exploiting nothing: the problem is tiny
  const int m = n_factors * n;
  std::vector<double> M((size_t)m * n);
  for (int f = 0; f < n_factors; ++f)
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) M[(size_t)j * m + f * n + i] = i <= j ? R_stack[(size_t)f * n * n + (size_t)j * n + i] : 0.0;
  for (int k = 0; k < n; ++k)
  {
    double sigma = 0.0;
    for (int i = k + 1; i < m; ++i) sigma += M[(size_t)k * m + i] * M[(size_t)k * m + i];
    const double alpha = M[(size_t)k * m + k];
    if (!(sigma > 1e-280)) continue;  // zero, or residue about to underflow (rdyn_tsqr.hip: tsqr_fold2d)
    const double norm = std::sqrt(alpha * alpha + sigma);
    const double beta = alpha > 0 ? -norm : norm;
    const double v0 = alpha - beta;
    const double scale = 2.0 / (v0 * v0 + sigma);
    for (int j = k + 1; j < n; ++j)
    {
      double w = v0 * M[(size_t)j * m + k];
      for (int i = k + 1; i < m; ++i) w += M[(size_t)k * m + i] * M[(size_t)j * m + i];
      const double f = scale * w;
      M[(size_t)j * m + k] -= f * v0;
      for (int i = k + 1; i < m; ++i) M[(size_t)j * m + i] -= f * M[(size_t)k * m + i];
    }
    M[(size_t)k * m + k] = beta;
    for (int i = k + 1; i < m; ++i) M[(size_t)k * m + i] = 0.0;
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) R_out[(size_t)j * n + i] = i <= j ? M[(size_t)j * m + i] : 0.0;
  return RDYN_OK;
}

}  // extern "C"
