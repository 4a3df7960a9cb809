"""GPU parity of the batched local inverse kinematics vs the C oracle's restatement of
computeLocalIk / computeWeigthedLocalIk (primitives_impl.h:1398-1468)."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu


def _setup(urdf, base, tool, N, spread, seed=7):
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import uniform_pm1
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool), OracleChain(path, base, tool)
    lo, hi = np.array(ref.spec.q_min), np.array(ref.spec.q_max)
    q_goal = np.clip(uniform_pm1(seed, (N, ref.n)), lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo))   # reachable goals
    seeds = np.clip(q_goal + spread * uniform_pm1(seed + 1, (N, ref.n)), lo, hi)
    T = ref.fk(q_goal)[:, -1]                        # (N, 3, 4)
    return chain, ref, q_goal, seeds, T


def _run(torch, chain, T, seeds, layout, **kw):
    Tt = np.ascontiguousarray(T.transpose(0, 2, 1))   # (N, 4, 3): the getTransformation record
    if layout == "element":
        tT = torch.from_numpy(np.ascontiguousarray(np.moveaxis(Tt, 0, -1))).cuda()
        ts = torch.from_numpy(np.ascontiguousarray(seeds.T)).cuda()
    else:
        tT, ts = torch.from_numpy(Tt).cuda(), torch.from_numpy(np.ascontiguousarray(seeds)).cuda()
    sol, st, it = chain.computeLocalIk(tT, ts, layout=layout, **kw)
    sol = sol.cpu().numpy()
    return (sol.T if layout == "element" else sol), st.cpu().numpy(), it.cpu().numpy()


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "tool0"), ("mixed_joints.urdf", "world", "tip"),
                                            ("planar_2r.urdf", "base", "l2")])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_local_ik_matches_oracle(urdf, base, tool, layout):
    torch = pytest.importorskip("torch")
    chain, ref, q_goal, seeds, T = _setup(urdf, base, tool, 1000, 0.25)
    sol, st, it = _run(torch, chain, T, seeds, layout, toll=1e-6, max_iterations=30)
    rsol, rst, rit = ref.local_ik(T, seeds, toll=1e-6, max_iter=30)
    # Undamped Gauss-Newton is chaotic on the poses it does not settle on quickly (the iterates cross singular
    # configurations; a rounding difference changes the path), so the pose-by-pose comparison is over the poses the oracle
    # solves within 8 updates; the rest must agree statistically.
    conv = (rst == 1) & (rit <= 8)
    assert conv.mean() > 0.8, conv.mean()
    assert (st[conv] == 1).all()
    assert np.array_equal(it[conv], rit[conv])
    assert (st != rst).mean() < 0.05, (st != rst).mean()
    # converged poses: same solution; Gauss-Newton contracts rounding differences, 1e-9 leaves room for cond(J'J)
    assert np.abs(sol[conv] - rsol[conv]).max() < 1e-9
    # ... and it is a solution: the pose error of the HIP result, measured by the oracle, is below the tolerance
    Ts = ref.fk(sol[conv])[:, -1]
    from oracle.oracle import frame_distance
    worst = max(np.linalg.norm(frame_distance(a, b)) for a, b in zip(T[conv][:200], Ts[:200]))
    assert worst < 1e-6


@pytest.mark.parametrize("urdf,base,tool,damping", [("ur10_like.urdf", "base_link", "tool0", 0.0), ("panda_like.urdf", "link0", "link8", 1e-3)])
def test_ik_resume_launches_are_bit_identical_to_one_launch(urdf, base, tool, damping):
    """The staged path (8 updates for every pose, then the survivors re-packed into dense waves and resumed by
    k_local_ik_resume) against ONE launch that runs every pose to the cap (the C-ABI takes that path when `iterations` is
    NULL): the same arithmetic per pose, so solution AND status must agree bit for bit on every pose -- including the
    slow, chaotic ones the oracle comparison above can only check statistically."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from rosdyn_amd._lib import lib, check
    chain, ref, q_goal, seeds, T = _setup(urdf, base, tool, 5000, 0.6, seed=21)   # wide seeds: many poses need > 8 updates
    Tt = torch.from_numpy(np.ascontiguousarray(T.transpose(0, 2, 1))).cuda()
    ts = torch.from_numpy(np.ascontiguousarray(seeds)).cuda()
    sol_a, st_a, it_a = chain.computeLocalIk(Tt, ts, toll=1e-8, max_iterations=40, damping=damping)
    b, N, lay = chain._batch("sample", ts)
    sol_b = torch.empty_like(sol_a)
    st_b = torch.empty(N, dtype=torch.int32, device="cuda")
    check(lib().rdyn_local_ik_damped(chain._h, C.byref(b), Tt.data_ptr(), None, 1e-8, float(damping), 40, sol_b.data_ptr(), st_b.data_ptr(), None))
    torch.cuda.synchronize()
    it = it_a.cpu().numpy()
    assert (it > 8).mean() > 0.05, (it > 8).mean()          # the resume launch really had work to do
    assert torch.equal(st_a, st_b)
    assert torch.equal(sol_a, sol_b)


def test_weighted_ik_and_joint_limits():
    """Position-only weights (orientation free) and targets whose unconstrained solution leaves the joint range: the
    bound-constrained QP must keep every iterate inside [q_min, q_max] exactly as the oracle's Goldfarb-Idnani does."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import uniform_pm1
    chain, ref, q_goal, seeds, T = _setup("panda_like.urdf", "link0", "link7", 600, 0.2, seed=11)
    assert chain.setInputJointsName(chain.getMoveableJointNames()[:6])     # 6 of the 7 joints: J'WJ can be definite
    from oracle.oracle import OracleChain
    path = os.path.join(FIXTURES, "panda_like.urdf")
    ref = OracleChain(path, "link0", "link7", input_joint_names=ref.spec.moveable[:6])
    q_goal = q_goal[:, :6].copy()
    lo, hi = np.array(ref.spec.q_min), np.array(ref.spec.q_max)
    q_goal = np.clip(q_goal * 3.0, lo - 0.3, hi + 0.3)                     # some goals outside the limits
    T = ref.fk(q_goal)[:, -1]
    seeds = np.clip(q_goal + 0.2 * uniform_pm1(5, q_goal.shape), lo, hi)
    w = [1.0, 1.0, 1.0, 0.5, 0.5, 0.5]
    Tt = torch.from_numpy(np.ascontiguousarray(T.transpose(0, 2, 1))).cuda()
    sol, st, it = chain.computeWeigthedLocalIk(Tt, w, torch.from_numpy(seeds).cuda(), toll=1e-6, max_iterations=12)
    sol, st, it = sol.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    rsol, rst, rit = ref.local_ik(T, seeds, weight=w, toll=1e-6, max_iter=12)
    assert (st != rst).mean() < 0.05, (st != rst).mean()
    ok = st >= 0
    assert (sol[ok] >= lo - 1e-12).all() and (sol[ok] <= hi + 1e-12).all()
    assert ((np.abs(sol - lo) < 1e-12) | (np.abs(sol - hi) < 1e-12)).any(), "test must exercise active bounds"
    conv = (rst == 1) & (rit <= 8)
    assert conv.sum() > 50 and (st[conv] == 1).all() and np.array_equal(it[conv], rit[conv])
    assert np.abs(sol[conv] - rsol[conv]).max() < 1e-8
    # poses that end on their bounds without reaching the target ran the same 12 bounded updates
    nc = (rst == 0) & (st == 0)
    assert np.median(np.abs(sol[nc] - rsol[nc]).max(axis=1)) < 1e-8


def test_ik_reports_singular_normal_matrix():
    """7 input joints: J'J is 7x7 of rank <= 6 -- status -1 for every pose, like the oracle; sol = the seed."""
    torch = pytest.importorskip("torch")
    chain, ref, q_goal, seeds, T = _setup("panda_like.urdf", "link0", "hand", 64, 0.1)
    sol, st, it = _run(torch, chain, T, seeds, "sample", toll=1e-9, max_iterations=5)
    rsol, rst, rit = ref.local_ik(T, seeds, toll=1e-9, max_iter=5)
    # the pivot that reveals the deficiency is rounding noise / (null-vector component)^2: a few poses may slip the floor
    assert (rst == -1).mean() > 0.9 and (st == -1).mean() > 0.9
    both = (rst == -1) & (st == -1)
    assert (it[both] == 0).all() and (rit[both] == 0).all() and np.array_equal(sol[both], seeds[both])


def test_get_multiplicity_matches_reference_enumeration():
    from rosdyn_amd import Chain
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0")
    q = np.array([0.1, -0.2, 0.3, 6.0, -6.0, 0.0])
    m = chain.getMultiplicity(q)
    qmax, qmin = chain.getQMax(), chain.getQMin()
    per_axis = [1 + int(np.floor((qmax[i] - q[i]) / (2 * np.pi))) + int(np.floor((q[i] - qmin[i]) / (2 * np.pi))) for i in range(6)]
    assert len(m) == int(np.prod(per_axis)) and np.array_equal(m[0], q)
    for v in m:
        assert (v <= qmax).all() and (v >= qmin).all()
        assert np.allclose(((v - q) / (2 * np.pi)) - np.round((v - q) / (2 * np.pi)), 0, atol=1e-12)


def test_damped_ik_makes_the_seven_dof_arm_solvable():
    """rdyn_local_ik_damped: with a Levenberg term the 7-DOF Panda-like arm (singular J'J) converges; pose-by-pose
    agreement with the oracle's damped loop on the poses it solves quickly."""
    torch = pytest.importorskip("torch")
    chain, ref, q_goal, seeds, T = _setup("panda_like.urdf", "link0", "hand", 1500, 0.2, seed=21)
    sol, st, it = _run(torch, chain, T, seeds, "sample", toll=1e-6, max_iterations=40, damping=1e-3)
    rsol, rst, rit = ref.local_ik(T, seeds, toll=1e-6, max_iter=40, damping=1e-3)
    assert (st == 1).mean() > 0.9 and (rst == 1).mean() > 0.9
    conv = (rst == 1) & (rit <= 10)
    assert conv.mean() > 0.5 and (st[conv] == 1).all()
    assert (np.abs(it[conv] - rit[conv]) <= 1).all()
    # a redundant arm: the damped step is unique, so the iterates (not only the reached pose) agree
    assert np.abs(sol[conv] - rsol[conv]).max() < 1e-6
    from oracle.oracle import frame_distance
    Ts = ref.fk(sol[st == 1][:200])[:, -1]
    assert max(np.linalg.norm(frame_distance(a, b)) for a, b in zip(T[st == 1][:200], Ts)) < 1e-6
    lo, hi = np.array(ref.spec.q_min), np.array(ref.spec.q_max)
    assert (sol[st >= 0] >= lo - 1e-12).all() and (sol[st >= 0] <= hi + 1e-12).all()


@pytest.mark.parametrize("layout", ["sample", "element"])
def test_frame_distance_family_matches_oracle(layout):
    """rdyn_frame_distance: getFrameDistance / getFrameDistanceQuat / getFrameDistanceQuatJac (frame_distance.h) on random
    frame pairs, including tiny relative rotations and rotations close to half a turn (trace < 0 branches)."""
    torch = pytest.importorskip("torch")
    from scipy.spatial.transform import Rotation
    from oracle.oracle import frame_distance, frame_distance_quat
    from rosdyn_amd.frames import getFrameDistance, getFrameDistanceQuat, getFrameDistanceQuatJac
    rng = np.random.default_rng(11)
    N = 600
    Ra = Rotation.random(N, random_state=1)
    rel = Rotation.random(N, random_state=2).as_rotvec()
    rel[0:100] *= 1e-9 / np.linalg.norm(rel[0:100], axis=1, keepdims=True)                       # tiny angles
    rel[100:250] *= (np.pi - rng.uniform(1e-9, 1e-3, 150))[:, None] / np.linalg.norm(rel[100:250], axis=1, keepdims=True)
    Rb = Ra * Rotation.from_rotvec(rel)
    Ta = np.concatenate([Ra.as_matrix(), rng.normal(size=(N, 3, 1))], axis=2)                    # (N, 3, 4)
    Tb = np.concatenate([Rb.as_matrix(), rng.normal(size=(N, 3, 1))], axis=2)

    def dev(T):
        rec = np.ascontiguousarray(T.transpose(0, 2, 1))                                         # (N, 4, 3): columns of [R | p]
        return torch.from_numpy(np.ascontiguousarray(np.moveaxis(rec, 0, -1)) if layout == "element" else rec).cuda()

    def host(t):
        return np.moveaxis(t.cpu().numpy(), -1, 0) if layout == "element" else t.cpu().numpy()
    ta, tb = dev(Ta), dev(Tb)
    d0 = host(getFrameDistance(ta, tb, layout=layout))
    d1 = host(getFrameDistanceQuat(ta, tb, layout=layout))
    d2, J2 = getFrameDistanceQuatJac(ta, tb, layout=layout)
    d2, J2 = host(d2), host(J2)
    r0 = np.array([frame_distance(a, b) for a, b in zip(Ta, Tb)])
    r1 = np.array([frame_distance_quat(a, b) for a, b in zip(Ta, Tb)])
    r2 = [frame_distance_quat(a, b, jac=True) for a, b in zip(Ta, Tb)]
    # near half a turn the angle-axis vector is ill conditioned in the matrix (d angle / d R ~ 1 / sin): 1e-9 there, 1e-12 elsewhere
    easy = np.ones(N, dtype=bool)
    easy[100:250] = False
    assert np.abs(d0[easy] - r0[easy]).max() < 1e-12 and np.abs(d0 - r0).max() < 1e-8
    assert np.abs(d1 - r1).max() < 1e-12
    assert np.abs(d2 - np.array([x[0] for x in r2])).max() < 1e-12
    assert np.abs(J2.transpose(0, 2, 1) - np.array([x[1] for x in r2])).max() < 1e-12             # column-major 6 x 6 per pair
    assert np.array_equal(d2[:, :3], -d1[:, :3]) and np.array_equal(d2[:, 3:], d1[:, 3:])       # :115 vs :75
