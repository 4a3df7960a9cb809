"""Host-side pieces of the normal-equation path (no GPU): the rank-revealing R factor and the base-parameter solve,
on the oracle's regressor of a real chain (structurally rank deficient)."""
import os

import numpy as np

from conftest import FIXTURES
from oracle.oracle import OracleChain
from rosdyn_amd.gram import r_factor, residual_sum_of_squares, solve_base_parameters
from rosdyn_amd.samples import trajectory_batch


def _stacked_regressor():
    ref = OracleChain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0", (0.0, 0.0, -9.806))
    q, dq, ddq = trajectory_batch(77, 400, ref.n)
    Y = ref.regressor(q, dq, ddq)                                  # (N, n, P)
    A = Y.reshape(-1, ref.P)                                       # rows (s, j)
    pi = np.empty(ref.P)
    from oracle.oracle import lib, _p
    lib().orc_nominal_parameters(ref._h, _p(pi))
    return A, pi


def test_r_factor_is_the_qr_factor_of_the_stacked_regressor():
    A, _ = _stacked_regressor()
    G = A.T @ A
    R, perm, rank = r_factor(G, rtol=1e-12)
    assert rank == np.linalg.matrix_rank(A, tol=1e-8 * np.linalg.norm(A, 2)) < A.shape[1]      # unobservable parameters exist
    # R'R reproduces the permuted Gram
    Gp = G[np.ix_(perm, perm)]
    assert np.abs(R.T @ R - Gp).max() <= 1e-9 * np.abs(G).max()
    # and R is the triangular factor Householder QR gives for the same column order (up to row signs)
    Rq = np.linalg.qr(A[:, perm[:rank]], mode="r")
    sgn = np.sign(np.diag(Rq))
    assert np.abs(R[:, :rank] - sgn[:, None] * Rq).max() <= 1e-7 * np.abs(Rq).max()
    assert (np.diag(R[:, :rank]) > 0).all() and np.allclose(np.tril(R[:, :rank], -1), 0.0)


def test_base_parameter_solve_reproduces_the_torques():
    A, pi = _stacked_regressor()
    tau = A @ pi
    x, rank = solve_base_parameters(A.T @ A, A.T @ tau)
    assert rank < A.shape[1]
    assert np.abs(A @ x - tau).max() <= 1e-8 * np.abs(tau).max()   # same torques from the minimum-norm parameters
    assert np.linalg.norm(x) <= np.linalg.norm(pi) * (1 + 1e-9)


def test_residual_sum_of_squares_from_the_accumulators():
    A, pi = _stacked_regressor()
    rng = np.random.default_rng(5)
    b = A @ pi + 1e-2 * rng.normal(size=A.shape[0])
    G, c, bb = A.T @ A, A.T @ b, np.array([b @ b])
    x, _ = solve_base_parameters(G, c)
    rss = residual_sum_of_squares(G, c, bb, x)
    direct = float(np.sum((A @ x - b) ** 2))
    assert abs(rss - direct) <= 1e-9 * float(b @ b)
    assert 0.5e-4 * A.shape[0] < rss < 1.5e-4 * A.shape[0]        # ~ sigma^2 per row


def test_cabi_solves_match_numpy():
    """rdyn_solve_normal_equations / rdyn_gram_r_factor / rdyn_solve_r_factor (host C++ behind the C-ABI: what a C++ caller of the
    library uses) against numpy: eigen-truncated minimum-norm solve, pivoted Cholesky, and the SVD solve of a QR factor."""
    import ctypes as C
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import r_factor_abi, solve_normal_equations_abi
    A, pi = _stacked_regressor()
    tau = A @ pi
    G, c = A.T @ A, A.T @ tau
    x_np, rank_np = solve_base_parameters(G, c)
    x, rank = solve_normal_equations_abi(G, c)
    assert rank == rank_np
    assert np.abs(x - x_np).max() <= 1e-8 * np.abs(x_np).max()
    assert np.abs(A @ x - tau).max() <= 1e-8 * np.abs(tau).max()
    R_np, perm_np, rk_np = r_factor(G, rtol=1e-12)
    R, perm, rk = r_factor_abi(G, rtol=1e-12)
    assert rk == rk_np and (perm == perm_np).all()
    assert np.abs(R - R_np).max() <= 1e-10 * np.abs(R_np).max()
    # minimum-norm solution from a QR factor of [A | tau]: R1 x = d, never forming A'A
    Rq = np.linalg.qr(np.column_stack([A, tau]), mode="r")          # (P + 1) x (P + 1)
    P = A.shape[1]
    R1 = np.asfortranarray(Rq[:P, :P])
    d = np.ascontiguousarray(Rq[:P, P])
    xs = np.zeros(P)
    rks = C.c_int(0)
    dp = C.POINTER(C.c_double)
    st = lib().rdyn_solve_r_factor(R1.ctypes.data_as(dp), P, P, P, d.ctypes.data_as(dp), 1e-10, xs.ctypes.data_as(dp), C.byref(rks))
    assert st == 0 and rks.value == rank_np
    x_ls = np.linalg.lstsq(A, tau, rcond=1e-10)[0]
    assert np.abs(xs - x_ls).max() <= 1e-8 * np.abs(x_ls).max()
    assert np.abs(A @ xs - tau).max() <= 1e-9 * np.abs(tau).max()


def test_r_factor_solve_survives_a_condition_number_the_gram_cannot():
    """cond(A) = 1e9: A'A has cond 1e18 > 1 / eps, the normal equations lose the small directions, the QR route keeps them."""
    import ctypes as C
    from rosdyn_amd._lib import lib
    rng = np.random.default_rng(11)
    m, n = 600, 12
    U, _ = np.linalg.qr(rng.normal(size=(m, n)))
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    sv = np.logspace(0, -9, n)
    A = U @ np.diag(sv) @ V.T
    x_true = V @ (rng.normal(size=n))
    b = A @ x_true
    R = np.asfortranarray(np.linalg.qr(np.column_stack([A, b]), mode="r"))
    xs = np.zeros(n)
    rk = C.c_int(0)
    dp = C.POINTER(C.c_double)
    d = np.ascontiguousarray(R[:n, n])
    R1 = np.asfortranarray(R[:n, :n])
    assert lib().rdyn_solve_r_factor(R1.ctypes.data_as(dp), n, n, n, d.ctypes.data_as(dp), 1e-13, xs.ctypes.data_as(dp), C.byref(rk)) == 0
    assert rk.value == n
    assert np.abs(xs - x_true).max() <= 1e-5 * np.abs(x_true).max()           # cond * eps = 1e-7 relative
    x_ne, rank_ne = solve_base_parameters(A.T @ A, A.T @ b, rtol=1e-15)
    assert np.abs(x_ne - x_true).max() > 1e-3 * np.abs(x_true).max()          # the Gram route has lost the small directions
