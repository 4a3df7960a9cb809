"""CPU checks of the IK restatement in oracle/ik_oracle.c (test infrastructure): the Goldfarb-Idnani QP against an
exhaustive active-set enumeration and its KKT conditions, getFrameDistance against scipy's rotation vector, and the
computeLocalIk loop (primitives_impl.h:1398-1468) on reachable targets."""
import itertools
import os

import numpy as np
import pytest

from conftest import FIXTURES
from oracle import oracle as O
from rosdyn_amd.samples import uniform_pm1


def _brute_force_box_qp(G, g0, lo, hi):
    n, best = len(g0), None
    for state in itertools.product((0, 1, 2), repeat=n):      # free / at lower / at upper
        F = [i for i in range(n) if state[i] == 0]
        A = [i for i in range(n) if state[i]]
        x = np.zeros(n)
        for i in A:
            x[i] = lo[i] if state[i] == 1 else hi[i]
        if F:
            x[F] = np.linalg.solve(G[np.ix_(F, F)], -g0[F] - G[np.ix_(F, A)] @ x[A])
        if (x >= lo - 1e-12).all() and (x <= hi + 1e-12).all():
            v = 0.5 * x @ G @ x + g0 @ x
            if best is None or v < best[0]:
                best = (v, x)
    return best[1]


def test_quadprog_matches_enumeration_and_kkt():
    rng = np.random.default_rng(2024)
    for trial in range(150):
        n = int(rng.integers(1, 7))
        Jm = rng.normal(size=(6, n))
        G, g0 = Jm.T @ Jm + 1e-3 * np.eye(n), 3.0 * rng.normal(size=n)
        lo, hi = -rng.uniform(0.05, 1.5, n), rng.uniform(0.05, 1.5, n)
        if trial % 3 == 0:
            lo, hi = lo + 2.0, hi + 2.5                         # x = 0 infeasible: the dual method starts outside
        CI, ci0 = np.hstack([np.eye(n), -np.eye(n)]), np.concatenate([-lo, hi])
        st, x = O.solve_quadprog(G, g0, CI, ci0)
        assert st == 0
        assert np.abs(x - _brute_force_box_qp(G, g0, lo, hi)).max() < 1e-11
        # KKT: gradient = sum of non-negative multipliers on active bounds
        grad = G @ x + g0
        for i in range(n):
            if abs(x[i] - lo[i]) < 1e-10:
                assert grad[i] > -1e-9
            elif abs(x[i] - hi[i]) < 1e-10:
                assert grad[i] < 1e-9
            else:
                assert abs(grad[i]) < 1e-9


def test_quadprog_general_inequalities_and_failures():
    # a non-box problem (the routine is the general one the reference calls): min |x|^2/2 - x1 - x2, x1 + x2 <= 1
    st, x = O.solve_quadprog(np.eye(2), [-1.0, -1.0], np.array([[-1.0], [-1.0]]), [1.0])
    assert st == 0 and np.allclose(x, [0.5, 0.5], atol=1e-14)
    st, _ = O.solve_quadprog(np.array([[1.0, 1.0], [1.0, 1.0]]), [0.0, 0.0], np.zeros((2, 0)), [])
    assert st == -1                                             # singular G
    st, _ = O.solve_quadprog(np.eye(1), [0.0], np.array([[1.0, -1.0]]), [-2.0, 1.0])
    assert st == -2                                             # x >= 2 and x <= 1


def test_frame_distance_is_the_rotation_vector():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    for k in range(200):
        Ra, Rb = Rotation.random(random_state=int(rng.integers(1 << 30))), Rotation.random(random_state=int(rng.integers(1 << 30)))
        if k % 10 == 0:
            Rb = Ra * Rotation.from_rotvec(rng.normal(size=3) * 1e-9)      # tiny angle
        if k % 10 == 1:
            v = rng.normal(size=3)
            Rb = Ra * Rotation.from_rotvec(v / np.linalg.norm(v) * (np.pi - 1e-6))   # close to half a turn (trace < 0 branches)
        pa, pb = rng.normal(size=3), rng.normal(size=3)
        Ta, Tb = np.hstack([Ra.as_matrix(), pa[:, None]]), np.hstack([Rb.as_matrix(), pb[:, None]])
        d = O.frame_distance(Ta, Tb)
        expect = -Ra.as_matrix() @ (Ra.inv() * Rb).as_rotvec()            # frame_distance.h:47-48
        assert np.abs(d[:3] - (pa - pb)).max() == 0.0
        assert np.abs(d[3:] - expect).max() < 1e-9 * max(1.0, np.abs(expect).max()) + 1e-12
    assert np.array_equal(O.frame_distance(Ta, Ta), np.zeros(6))


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "tool0"), ("planar_2r.urdf", "base", "l2")])
def test_local_ik_reaches_reachable_targets(urdf, base, tool):
    ref = O.OracleChain(os.path.join(FIXTURES, urdf), base, tool)
    N = 64
    lo, hi = np.array(ref.spec.q_min), np.array(ref.spec.q_max)
    goal = np.clip(uniform_pm1(1, (N, ref.n)), lo, hi)
    seeds = np.clip(goal + 0.2 * uniform_pm1(2, (N, ref.n)), lo, hi)
    T = ref.fk(goal)[:, -1]
    sol, st, it = ref.local_ik(T, seeds, toll=1e-8, max_iter=40)
    ok = st == 1
    assert ok.mean() > 0.85
    Ts = ref.fk(sol[ok])[:, -1]
    assert max(np.linalg.norm(O.frame_distance(a, b)) for a, b in zip(T[ok], Ts)) < 1e-8
    assert (sol[ok] >= lo - 1e-12).all() and (sol[ok] <= hi + 1e-12).all()
    # already at the goal: converged with zero updates (primitives_impl.h:1408-1412)
    sol0, st0, it0 = ref.local_ik(T, goal, toll=1e-8, max_iter=40)
    assert (st0 == 1).all() and (it0 == 0).all() and np.array_equal(sol0, goal)
    # weighted variant with unit weights == unweighted (primitives_impl.h:1446-1452)
    solw, stw, itw = ref.local_ik(T, seeds, weight=np.ones(6), toll=1e-8, max_iter=40)
    assert np.array_equal(stw, st) and np.array_equal(solw, sol)


def test_frame_distance_quat_and_its_jacobian():
    """getFrameDistanceQuat / getFrameDistanceQuatJac (frame_distance.h:73-126) against scipy's quaternion of R_a' R_b."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(8)
    for k in range(100):
        Ra, Rb = Rotation.random(random_state=2 * k), Rotation.random(random_state=2 * k + 1)
        pa, pb = rng.normal(size=3), rng.normal(size=3)
        Ta, Tb = np.hstack([Ra.as_matrix(), pa[:, None]]), np.hstack([Rb.as_matrix(), pb[:, None]])
        q = (Ra.inv() * Rb).as_quat()                               # x, y, z, w
        if q[3] < 0:
            q = -q
        d = O.frame_distance_quat(Ta, Tb)
        assert np.abs(d[:3] - (pa - pb)).max() == 0.0
        assert np.abs(d[3:] - (-2.0 * Ra.as_matrix() @ q[:3])).max() < 1e-13
        dj, J = O.frame_distance_quat(Ta, Tb, jac=True)
        assert np.abs(dj[:3] - (pb - pa)).max() == 0.0 and np.array_equal(dj[3:], d[3:])
        v = q[:3]
        K = q[3] * np.eye(3) - np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        expect = np.eye(6)
        expect[3:, 3:] = Ra.as_matrix() @ K @ Ra.as_matrix().T
        assert np.abs(J - expect).max() < 1e-13
        # for small relative rotations the quaternion form tends to the angle-axis form
    Rb = Ra * Rotation.from_rotvec([1e-5, -2e-5, 3e-5])
    Tb = np.hstack([Rb.as_matrix(), pb[:, None]])
    assert np.abs(O.frame_distance_quat(Ta, Tb) - O.frame_distance(Ta, Tb)).max() < 1e-13
