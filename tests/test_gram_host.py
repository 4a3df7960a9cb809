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
