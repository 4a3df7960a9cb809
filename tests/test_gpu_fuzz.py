"""Seeded structural fuzz (-m gpu): random serial chains with every joint kind the reference maps (revolute, continuous,
prismatic, fixed, floating/planar -> fixed, primitives_impl.h:74-83), optional <origin>/<axis>/<inertial>/<limit>
elements, side branches that are not on the chain, random permuted subsets of input joints -- every batched entry point
against the CPU oracle.  The URDF text goes through two independent readers (rdyn_urdf.cpp vs oracle/urdf_model.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _fmt(v):
    return " ".join("%.17g" % float(x) for x in v)


# RDYN_FUZZ_OFFSET=k shifts the seeds of the fuzzed chains (extended campaigns: tools/gpu_suite.sh runs the committed seeds)
FUZZ_OFFSET = int(os.environ.get("RDYN_FUZZ_OFFSET", "0"))


def random_chain_xml(seed):
    rng = np.random.default_rng(seed)
    nj = int(rng.integers(1, 11))
    kinds = ["revolute", "continuous", "prismatic", "fixed", "floating", "planar"]
    probs = [0.45, 0.15, 0.15, 0.15, 0.05, 0.05]
    links, joints = ["<link name='L0'/>"], []
    n_moving = 0
    for i in range(nj):
        kind = str(rng.choice(kinds, p=probs))
        if i == nj - 1 and n_moving == 0:
            kind = "revolute"                                   # at least one moveable joint
        n_moving += kind in ("revolute", "continuous", "prismatic")
        origin = ""
        if rng.random() < 0.85:
            origin = "<origin xyz='%s' rpy='%s'/>" % (_fmt(0.3 * rng.uniform(-1, 1, 3)), _fmt(rng.uniform(-3.1, 3.1, 3)))
            if rng.random() < 0.15:
                origin = "<origin xyz='%s'/>" % _fmt(0.3 * rng.uniform(-1, 1, 3))       # rpy missing
        axis = ""
        if rng.random() < 0.8:
            axis = "<axis xyz='%s'/>" % _fmt(rng.uniform(-1, 1, 3) * rng.choice([1.0, 2.5]))   # not normalised
        limit = ""
        if kind in ("revolute", "prismatic") or rng.random() < 0.3:
            limit = "<limit lower='%.17g' upper='%.17g' effort='50' velocity='2'/>" % (-rng.uniform(1.5, 3), rng.uniform(1.5, 3))
        joints.append("<joint name='J%d' type='%s'><parent link='L%d'/><child link='L%d'/>%s%s%s</joint>"
                      % (i, kind, i, i + 1, origin, axis, limit))
        inertial = ""
        if rng.random() < 0.85:
            B = rng.normal(size=(3, 3))
            I = B @ B.T * 0.01 + 0.01 * np.eye(3)
            io = "<origin xyz='%s' rpy='%s'/>" % (_fmt(0.1 * rng.uniform(-1, 1, 3)), _fmt(rng.uniform(-3, 3, 3))) if rng.random() < 0.8 else ""
            inertial = ("<inertial>%s<mass value='%.17g'/><inertia ixx='%.17g' ixy='%.17g' ixz='%.17g' iyy='%.17g' iyz='%.17g' "
                        "izz='%.17g'/></inertial>" % (io, rng.uniform(0.2, 5), I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        links.append("<link name='L%d'>%s</link>" % (i + 1, inertial))
        if rng.random() < 0.25:                                  # a side branch hanging off the chain
            links.append("<link name='S%d'><inertial><mass value='1'/><inertia ixx='1' ixy='0' ixz='0' iyy='1' iyz='0' izz='1'/></inertial></link>" % i)
            joints.append("<joint name='B%d' type='revolute'><parent link='L%d'/><child link='S%d'/><axis xyz='0 1 0'/>"
                          "<limit lower='-1' upper='1' effort='1' velocity='1'/></joint>" % (i, i, i))
    order = rng.permutation(len(links) + len(joints))           # element order in the file must not matter
    elems = links + joints
    body = "".join(elems[k] for k in order)
    lo = int(rng.integers(0, max(1, nj // 3 + 1)))               # chain = a sub-path of the tree
    hi = nj
    return "<robot name='fuzz%d'>%s</robot>" % (seed, body), "L%d" % lo, "L%d" % hi, rng


@pytest.mark.parametrize("seed", range(64))
def test_fuzzed_chain_all_entry_points(seed):
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch, uniform_pm1
    xml, base, tool, rng = random_chain_xml(1000 + seed + FUZZ_OFFSET)
    grav = tuple(rng.uniform(-10, 10, 3))
    chain = Chain(xml, base, tool, grav)
    ref0 = OracleChain(xml, base, tool, grav)
    if ref0.n == 0:
        pytest.skip("sub-path without a moveable joint")
    inputs = None
    if seed % 2 == 1 and ref0.n >= 2:                            # permuted subset of the moveable joints
        names = list(ref0.spec.moveable)
        k = int(rng.integers(1, len(names) + 1))
        inputs = [names[i] for i in rng.permutation(len(names))[:k]]
        assert chain.setInputJointsName(inputs)
    ref = OracleChain(xml, base, tool, grav, input_joint_names=inputs)
    assert chain.getActiveJointsNumber() == ref.n and chain.getLinksNumber() == ref.L and chain.getLinksName() == ref.spec.link_names
    assert np.array_equal(chain.getQMax(), np.array(ref.spec.q_max)) and np.array_equal(chain.getQMin(), np.array(ref.spec.q_min))
    N, n, L, P = 257, ref.n, ref.L, ref.P
    q, dq, ddq, dddq = trajectory_batch(seed, N, n, order=4)
    ext = 3.0 * uniform_pm1(seed + 99, (N, L, 6))
    layout = "element" if seed % 3 == 0 else "sample"
    if layout == "element":
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()
        host = lambda t: np.moveaxis(t.cpu().numpy(), -1, 0)
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        host = lambda t: t.cpu().numpy()
    tq, tdq, tddq, tdddq, text = dev(q), dev(dq), dev(ddq), dev(dddq), dev(ext)

    def close(a, b, what, tol=TOL):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), "%s: %.3e" % (what, np.abs(a - b).max())

    pi = chain.getNominalParameters()
    close(host(chain.getTransformations(tq, layout=layout)).transpose(0, 1, 3, 2), ref.fk(q), "T")
    close(host(chain.getJacobian(tq, layout=layout)).transpose(0, 2, 1), ref.jacobian(q), "J")
    mid = chain.getLinksName()[L // 2]
    close(host(chain.getJacobianLink(tq, mid, layout=layout)).transpose(0, 2, 1), ref.jacobian_link(q, L // 2), "J link")
    close(host(chain.getTwist(tq, tdq, layout=layout)), ref.twist(q, dq), "twist")
    a, al, an = ref.dtwist(q, dq, ddq, parts=True)
    close(host(chain.getDTwist(tq, tdq, tddq, layout=layout)), a, "dtwist")
    close(host(chain.getDTwistLinearPart(tq, tddq, layout=layout)), al, "dtwist linear")
    close(host(chain.getDTwistNonLinearPart(tq, tdq, layout=layout)), an, "dtwist non-linear")
    close(host(chain.getDDTwist(tq, tdq, tddq, tdddq, layout=layout)), ref.ddtwist(q, dq, ddq, dddq), "jerk")
    jl, jn = ref.ddtwist_parts(q, dq, ddq, dddq)
    close(host(chain.getDDTwistLinearPart(tq, tdddq, layout=layout)), jl, "jerk linear")
    close(host(chain.getDDTwistNonLinearPart(tq, tdq, tddq, layout=layout)), jn, "jerk non-linear")
    _, wr = ref.joint_torque(q, dq, ddq, ext=ext, wrenches=True)
    close(host(chain.getWrench(tq, tdq, tddq, text, layout=layout)), wr, "wrenches")
    tau = ref.joint_torque(q, dq, ddq)
    close(host(chain.getJointTorque(tq, tdq, tddq, layout=layout)), tau, "tau")
    close(host(chain.getJointTorqueNonLinearPart(tq, tdq, layout=layout)), ref.joint_torque(q, dq, 0 * ddq), "tau nl")
    close(host(chain.getJointTorqueExt(tq, tdq, tddq, text, layout=layout)), ref.joint_torque(q, dq, ddq, ext=ext), "tau ext")
    close(host(chain.getJointInertia(tq, layout=layout)).transpose(0, 2, 1), ref.joint_inertia(q), "M")
    Yr = ref.regressor(q, dq, ddq)
    Y, tau2 = chain.getRegressor(tq, tdq, tddq, layout=layout, with_torque=True)
    close(host(Y).transpose(0, 2, 1), Yr, "Y")
    close(host(tau2), tau, "tau fused")
    close(np.einsum("snp,p->sn", Yr, pi), tau, "Y pi = tau (oracle, product parameters)", 1e-10)
    if layout == "sample":
        Ys = chain.getRegressor(tq, tdq, tddq, y_layout="stacked")
        close(Ys.cpu().numpy().reshape(P, N, n).transpose(1, 2, 0), Yr, "Y stacked")
    if P + 1 <= 111:
        eq, edq, eddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))
        etau = torch.from_numpy(np.ascontiguousarray(tau.T)).cuda()
        A = Yr.transpose(1, 0, 2).reshape(n * N, P)
        Gr, cr = A.T @ A, A.T @ tau.T.reshape(-1)
        # fused path (LDS tile or global image), then the two-kernel path with row blocks of 128, 100 and 7 rows (the
        # last two are not multiples of the 16-row MFMA group: groups straddle row blocks)
        for chunk in (0, 128, 100, 7):
            G, c, bb = chain.getRegressorGram(eq, edq, eddq, etau, layout="element", chunk_samples=chunk)
            assert np.linalg.norm(G.cpu().numpy() - Gr) <= 1e-10 * max(np.linalg.norm(Gr), 1e-300), ("G", chunk)
            assert np.linalg.norm(c.cpu().numpy() - cr) <= 1e-10 * max(np.linalg.norm(cr), 1e-300), ("c", chunk)
            assert abs(bb.item() - (tau ** 2).sum()) <= 1e-10 * max((tau ** 2).sum(), 1e-300)
        # the R factor of [A | tau] (round 3: chains with non-input joints are swept through their reduced companion and the factor
        # expanded by a small QR): R'R = [G c; c' bb] wherever the entry point serves the chain
        # (round 5: input joints in ANY order -- the sorted view is swept, every row's inputs read through its map)
        from rosdyn_amd._lib import lib
        if lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) > 0:
            R1 = chain.getRegressorTsqr(eq, edq, eddq, etau, layout="element").cpu().numpy()
            full = np.zeros((P + 1, P + 1))
            full[:P, :P], full[:P, P], full[P, :P], full[P, P] = Gr, cr, cr, (tau ** 2).sum()
            assert np.allclose(np.tril(R1, -1), 0.0)
            assert np.abs(R1.T @ R1 - full).max() <= 1e-10 * max(np.abs(full).max(), 1e-300), "R factor"
            # and a batch above the threshold of the preconditioned route (4 096 samples): random joint kinds, axes and inertias
            # through the pass-B kernel's general sweeper, the reduced companion wherever the chain has non-input joints
            N2 = 4200
            q2, dq2, ddq2 = trajectory_batch(seed + 500, N2, n)
            tau2 = ref.joint_torque(q2, dq2, ddq2) + 0.01 * np.random.default_rng(seed).normal(size=(N2, n))
            M2 = np.column_stack([ref.regressor(q2, dq2, ddq2).reshape(-1, P), tau2.reshape(-1)])
            R2 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q2, dq2, ddq2, tau2))).cpu().numpy()
            G2 = M2.T @ M2
            assert np.allclose(np.tril(R2, -1), 0.0)
            assert np.abs(R2.T @ R2 - G2).max() <= 1e-11 * max(np.abs(G2).max(), 1e-300), "R factor, preconditioned route"
            s_ref, s_gpu = np.linalg.svd(np.linalg.qr(M2, mode="r"), compute_uv=False), np.linalg.svd(R2, compute_uv=False)
            keep = s_ref > 1e-8 * s_ref[0]
            assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-7, "singular values, preconditioned route"
        else:
            assert not (2 <= n <= 7 and ref.nJ <= 8), "the factor entry points serve 2..8 input joints in any order"


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023])
def test_ragged_batch_sizes(N):
    """Batch sizes around every tiling boundary (16-sample LDS tiles, 64-lane waves, 256-thread blocks, row-pair lanes)."""
    torch = pytest.importorskip("torch")
    import os
    from conftest import FIXTURES
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "panda_like.urdf")
    grav = (0.0, 0.0, -9.806)
    chain, ref = Chain(path, "link0", "hand", grav), OracleChain(path, "link0", "hand", grav)
    n, P = ref.n, ref.P
    q, dq, ddq = trajectory_batch(N, N, n)
    Yr, tau = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    eq, edq, eddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))

    def close(a, b, what):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), what
    Y, t2 = chain.getRegressor(tq, tdq, tddq, with_torque=True)
    close(Y.cpu().numpy().transpose(0, 2, 1), Yr, "per-sample")
    close(t2.cpu().numpy(), tau, "tau")
    close(chain.getRegressor(tq, tdq, tddq, y_layout="stacked").cpu().numpy().reshape(P, N, n).transpose(1, 2, 0), Yr, "stacked")
    close(chain.getRegressor(eq, edq, eddq, layout="element").cpu().numpy().transpose(2, 1, 0), Yr, "element")
    close(chain.getJointTorque(tq, tdq, tddq).cpu().numpy(), tau, "torque")
    close(chain.getJointInertia(tq).cpu().numpy().transpose(0, 2, 1), ref.joint_inertia(q), "M")
    etau = torch.from_numpy(np.ascontiguousarray(tau.T)).cuda()
    A = Yr.transpose(1, 0, 2).reshape(n * N, P)
    Gr = A.T @ A
    for chunk in (0, 64):
        G, c, bb = chain.getRegressorGram(eq, edq, eddq, etau, layout="element", chunk_samples=chunk)
        assert np.linalg.norm(G.cpu().numpy() - Gr) <= 1e-10 * np.linalg.norm(Gr), chunk
    T = chain.getTransformation(tq)
    sol, st, it = chain.computeLocalIk(T, tq, toll=1e-9, max_iterations=3)      # seeds = goals: converged, untouched
    assert (st.cpu().numpy() == -1).all() or ((st.cpu().numpy() == 1).all() and torch.equal(sol, tq))


def test_mixed_plan_with_fuzzed_chains_and_ragged_items():
    """rdyn_multi_plan over random chains of different joint counts, items of different (also zero) sample counts."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.samples import trajectory_batch
    chains, refs, batches = [], [], []
    for k, seed in enumerate(range(2000, 2012)):
        xml, base, tool, rng = random_chain_xml(seed)
        grav = (0.0, 0.0, -9.806)
        ref = OracleChain(xml, base, tool, grav)
        if ref.n == 0:
            continue
        chains.append(Chain(xml, base, tool, grav))
        refs.append(ref)
        N = [0, 1, 77, 256, 300, 513][k % 6]
        batches.append(trajectory_batch(seed, max(N, 1), ref.n) if N else tuple(np.zeros((0, ref.n)) for _ in range(3)))
    dev = [tuple(torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in b) for b in batches]
    mc = MultiChainRegressor([(c,) + d for c, d in zip(chains, dev)])
    mc.run()
    torch.cuda.synchronize()
    for ref, b, Y, tau in zip(refs, batches, mc.Y, mc.tau):
        if b[0].shape[0] == 0:
            continue
        Yr = ref.regressor(*b)
        assert np.abs(Y.cpu().numpy().transpose(2, 1, 0) - Yr).max() <= TOL * max(1.0, np.abs(Yr).max())
        tr = ref.joint_torque(*b)
        assert np.abs(tau.cpu().numpy().T - tr).max() <= TOL * max(1.0, np.abs(tr).max())


@pytest.mark.parametrize("scale_q,scale_dq,scale_ddq", [(1e3, 1.0, 1.0), (3.0, 50.0, 500.0), (1e-9, 1e-9, 1e-9), (1e5, 20.0, 100.0)])
def test_extreme_input_ranges(scale_q, scale_dq, scale_ddq):
    """Large joint angles (argument reduction of sincos up to 1e5 rad), fast motions and vanishing inputs: same relative
    tolerance as everywhere else, |delta| <= 1e-11 max(1, |ref|_inf)."""
    torch = pytest.importorskip("torch")
    import os
    from conftest import FIXTURES
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "mixed_joints.urdf")
    grav = (0.1, -0.2, -9.806)
    chain, ref = Chain(path, "world", "tip", grav), OracleChain(path, "world", "tip", grav)
    N, n = 513, ref.n
    q, dq, ddq = trajectory_batch(17, N, n)
    q, dq, ddq = q * scale_q, dq * scale_dq, ddq * scale_ddq
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))

    def close(a, b, what):
        assert np.abs(a - b).max() <= TOL * max(1.0, np.abs(b).max()), "%s: %.3e vs scale %.3e" % (what, np.abs(a - b).max(), np.abs(b).max())
    Y, tau = chain.getRegressor(tq, tdq, tddq, with_torque=True)
    close(Y.cpu().numpy().transpose(0, 2, 1), ref.regressor(q, dq, ddq), "Y")
    close(tau.cpu().numpy(), ref.joint_torque(q, dq, ddq), "tau fused")
    close(chain.getJointTorque(tq, tdq, tddq).cpu().numpy(), ref.joint_torque(q, dq, ddq), "tau")
    close(chain.getJointInertia(tq).cpu().numpy().transpose(0, 2, 1), ref.joint_inertia(q), "M")
    close(chain.getTransformations(tq).cpu().numpy().transpose(0, 1, 3, 2), ref.fk(q), "T")
    close(chain.getDTwist(tq, tdq, tddq).cpu().numpy(), ref.dtwist(q, dq, ddq), "dtwist")


def random_long_chain_xml(seed):
    """2..16 chain joints, at most 7 of them moving (revolute / continuous / prismatic), fixed frames anywhere -- head, middle, tail --
    so that chains LONGER than the kernels sweep (> 10 joints) and every reduced-companion shape occur."""
    rng = np.random.default_rng(seed)
    nj = int(rng.integers(2, 17))
    n_move = int(rng.integers(2, min(7, nj) + 1))
    moving = set(rng.choice(nj, size=n_move, replace=False).tolist())
    links, joints = ["<link name='L0'/>"], []
    for i in range(nj):
        kind = str(rng.choice(["revolute", "continuous", "prismatic"], p=[0.6, 0.2, 0.2])) if i in moving else str(rng.choice(["fixed", "floating", "planar"], p=[0.8, 0.1, 0.1]))
        origin = "<origin xyz='%s' rpy='%s'/>" % (_fmt(0.3 * rng.uniform(-1, 1, 3)), _fmt(rng.uniform(-3.1, 3.1, 3))) if rng.random() < 0.9 else ""
        axis = "<axis xyz='%s'/>" % _fmt(rng.uniform(-1, 1, 3)) if rng.random() < 0.85 else ""
        limit = "<limit lower='-2' upper='2' effort='50' velocity='2'/>" if kind in ("revolute", "prismatic") else ""
        joints.append("<joint name='J%d' type='%s'><parent link='L%d'/><child link='L%d'/>%s%s%s</joint>" % (i, kind, i, i + 1, origin, axis, limit))
        inertial = ""
        if rng.random() < 0.8:
            B = rng.normal(size=(3, 3))
            I = B @ B.T * 0.01 + 0.01 * np.eye(3)
            io = "<origin xyz='%s' rpy='%s'/>" % (_fmt(0.1 * rng.uniform(-1, 1, 3)), _fmt(rng.uniform(-3, 3, 3)))
            inertial = ("<inertial>%s<mass value='%.17g'/><inertia ixx='%.17g' ixy='%.17g' ixz='%.17g' iyy='%.17g' iyz='%.17g' "
                        "izz='%.17g'/></inertial>" % (io, rng.uniform(0.2, 5), I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        links.append("<link name='L%d'>%s</link>" % (i + 1, inertial))
    return "<robot name='long%d'>%s%s</robot>" % (seed, "".join(links), "".join(joints)), "L0", "L%d" % nj, rng


@pytest.mark.parametrize("seed", range(24))
def test_fuzzed_chain_identification_step(seed):
    """Round 4: the identification step on random chains of 2..16 joints (fixed frames in front, in the middle, behind; prismatic joints;
    more than 10 joints) with random friction / spring components: dense regressor, torque, normal equations and the R factor of
    [Y | C | tau] against the oracle's rows -- through every route the shape and the batch size select (register / LDS-resident
    Householder folds below 4 096 samples, preconditioned CholeskyQR above; reduced companion + expansion wherever a joint is fixed)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.components import ComponentSet
    from rosdyn_amd.samples import trajectory_batch
    xml, base, tool, rng = random_long_chain_xml(5000 + seed + FUZZ_OFFSET)
    grav = tuple(rng.uniform(-10, 10, 3))
    chain, ref = Chain(xml, base, tool, grav), OracleChain(xml, base, tool, grav)
    n, P = ref.n, ref.P
    N = 300 if seed % 2 == 0 else 5000
    q, dq, ddq = trajectory_batch(seed, N, n)
    n_c = int(rng.integers(1, n + 1))
    specs = []
    for j in rng.permutation(n)[:n_c]:
        ty = int(rng.integers(0, 3))
        specs.append((ty, int(j), 1e-3, 5.0, [0.4, 0.9, 0.03][:3 if ty == 1 else 2]))
    comps = ComponentSet([dict(type=s[0], joint=s[1], min_velocity=s[2], max_velocity=s[3], parameters=s[4]) for s in specs], n)
    K = comps.columns
    Cm, tau_c = components_regressor(specs, n, q, dq)
    Yr, tr = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq)
    tau = tr + tau_c + 1e-3 * rng.normal(size=(N, n))
    M = np.column_stack([Yr.reshape(-1, P), Cm.reshape(N * n, K), tau.reshape(-1)])
    G = M.T @ M
    tq, tdq, tddq, ttau = (torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))
    Y, tg = chain.getRegressor(tq, tdq, tddq, with_torque=True)
    assert np.abs(Y.cpu().numpy().transpose(0, 2, 1) - Yr).max() <= TOL * max(1.0, np.abs(Yr).max())
    assert np.abs(tg.cpu().numpy() - tr).max() <= TOL * max(1.0, np.abs(tr).max())
    Ys = chain.getRegressor(tq, tdq, tddq, y_layout="stacked").cpu().numpy().reshape(P, N, n).transpose(1, 2, 0)
    assert np.abs(Ys - Yr).max() <= TOL * max(1.0, np.abs(Yr).max())   # (long chains: the LDS-staged expanding sweep, odd row counts too)
    assert np.abs(chain.getJointTorque(tq, tdq, tddq).cpu().numpy() - tr).max() <= TOL * max(1.0, np.abs(tr).max())
    Gg, cg, bbg = chain.getIdentificationGram(comps, tq, tdq, tddq, ttau)
    C = P + K
    full = np.zeros((C + 1, C + 1))
    full[:C, :C], full[:C, C], full[C, :C], full[C, C] = Gg.cpu().numpy(), cg.cpu().numpy(), cg.cpu().numpy(), float(bbg.item())
    assert np.linalg.norm(full - G) <= 1e-10 * np.linalg.norm(G)
    R1 = chain.getIdentificationTsqr(comps, tq, tdq, tddq, ttau).cpu().numpy()
    assert R1.shape == (C + 1, C + 1) and np.allclose(np.tril(R1, -1), 0.0)
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False) if M.shape[0] >= C + 1 else None
    if s_ref is not None:
        s_gpu = np.linalg.svd(R1, compute_uv=False)
        keep = s_ref > 1e-8 * s_ref[0]
        assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-8
