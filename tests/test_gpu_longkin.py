"""GPU parity of the by-link kinematic outputs on chains LONGER than the unrolled kernels sweep (11 .. 32 chain joints): the
run-time-length kernels of rdyn_long_kin.hip against the C oracle.  The reference's default build has no bound on the chain
length (rosdyn_core/CMakeLists.txt:12-16); getTransformation(s) / getJacobian / getJacobianLink / getTwist / getDTwist
(primitives_impl.h:863-1027, 1082-1124), the split and jerk sweeps (:1029-1080, 1126-1223), getWrench and getJointTorque with
external wrenches (:1225-1272) and computeLocalIk (:1398-1433) are served for such chains with input joints in ANY order
(setInputJointsName, :705-737)."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.2, -0.3, -9.7)
TOL = 1e-11


def _close(a, b, what):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(np.asarray(a) - b).max())
    assert err <= TOL * scale, "%s: %.3e > %.1e * %.3g" % (what, err, TOL, scale)


def generated_long_chain(nj, seed):
    """nj chain joints: every third one fixed, every fifth prismatic, the rest revolute; links with and without inertial data."""
    from rosdyn_amd.samples import uniform_pm1
    r = uniform_pm1(seed, (nj + 1, 16))
    links = ["<link name='l0'/>"]
    joints = []
    for i in range(nj):
        k = r[i]
        typ = "fixed" if i % 3 == 2 else ("prismatic" if i % 5 == 3 else "revolute")
        joints.append(
            "<joint name='j%d' type='%s'><parent link='l%d'/><child link='l%d'/>"
            "<origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g %.17g'/><axis xyz='%.17g %.17g %.17g'/>"
            "<limit lower='-3' upper='3' effort='10' velocity='2'/></joint>"
            % (i, typ, i, i + 1, 0.1 * k[0], 0.1 * k[1], 0.08 + 0.05 * k[2], k[3], k[4], k[5], k[6], k[7], 1.0 + 0.5 * k[8]))
        if i % 7 == 4:
            links.append("<link name='l%d'/>" % (i + 1))
        else:
            links.append(
                "<link name='l%d'><inertial><origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g 0'/><mass value='%.17g'/>"
                "<inertia ixx='%.17g' ixy='%.17g' ixz='%.17g' iyy='%.17g' iyz='%.17g' izz='%.17g'/></inertial></link>"
                % (i + 1, 0.05 * k[9], 0.05 * k[10], 0.05 * k[11], 0.3 * k[12], 0.3 * k[13], 1.2 + k[14],
                   0.02, 0.002 * k[15], -0.001, 0.03, 0.0015, 0.025))
    return "<robot name='long%d'>%s%s</robot>" % (nj, "".join(links), "".join(joints))


def _cases():
    with open(os.path.join(FIXTURES, "ur10_public_long.urdf")) as f:
        ur = f.read()
    g20 = generated_long_chain(20, 2020)
    g32 = generated_long_chain(32, 3232)
    return {
        "ur10_long": (ur, "base_link", "tcp", None),                                        # 14 joints, 6 input joints
        "ur10_long_permuted": (ur, "base_link", "tcp", ["wrist_2_joint", "shoulder_pan_joint", "elbow_joint", "wrist_3_joint"]),
        "gen20_all": (g20, "l0", "l20", None),                                              # 14 input joints (> 10: rdyn_long_local.hip serves regressor / inertia)
        "gen20_permuted": (g20, "l0", "l20", ["j13", "j0", "j9", "j4", "j16", "j1", "j7"]),
        "gen32_all": (g32, "l0", "l32", None),                                              # the longest chain a build holds
    }


def _pair(case):
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    xml, base, tool, inputs = _cases()[case]
    chain, ref = Chain(xml, base, tool, GRAV), OracleChain(xml, base, tool, GRAV, input_joint_names=inputs)
    if inputs:
        assert chain.setInputJointsName(inputs)
    assert chain.getJointsNumber() > 10 and chain.getActiveJointsNumber() == ref.n
    return chain, ref


def _io(torch, layout):
    if layout == "element":
        return (lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()), (lambda t: np.moveaxis(t.cpu().numpy(), -1, 0))
    return (lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()), (lambda t: t.cpu().numpy())


@pytest.mark.parametrize("case", ["ur10_long", "ur10_long_permuted", "gen20_all", "gen20_permuted", "gen32_all"])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_long_chain_frames_jacobians_twists(case, layout):
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _pair(case)
    N, n = 601, ref.n
    q, dq, ddq = trajectory_batch(17, N, n)
    dev, host = _io(torch, layout)
    tq, tdq, tddq = dev(q), dev(dq), dev(ddq)
    T = ref.fk(q)
    _close(host(chain.getTransformations(tq, layout=layout)).transpose(0, 1, 3, 2), T, "T links")
    _close(host(chain.getTransformation(tq, layout=layout)).transpose(0, 2, 1), T[:, -1], "T tool")
    _close(host(chain.getJacobian(tq, layout=layout)).transpose(0, 2, 1), ref.jacobian(q), "J")
    tw = ref.twist(q, dq)
    _close(host(chain.getTwist(tq, tdq, layout=layout)), tw, "twists")
    _close(host(chain.getDTwist(tq, tdq, tddq, layout=layout)), ref.dtwist(q, dq, ddq), "dtwists")
    links = chain.getLinksName()
    for i in sorted(set([0, 1, len(links) // 2, len(links) - 2, len(links) - 1])):
        J = host(chain.getJacobianLink(tq, links[i], layout=layout)).transpose(0, 2, 1)
        _close(J, ref.jacobian_link(q, i), "J link %d" % i)
        _close(host(chain.getTransformationLink(tq, links[i], layout=layout)).transpose(0, 2, 1), T[:, i], "T link %d" % i)
        _close(host(chain.getTwistLink(tq, tdq, links[i], layout=layout)), tw[:, i], "twist link %d" % i)
    assert torch.equal(chain.getJacobianLink(tq, links[-1], layout=layout), chain.getJacobian(tq, layout=layout))


@pytest.mark.parametrize("case", ["ur10_long", "ur10_long_permuted", "gen20_all", "gen20_permuted", "gen32_all"])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_long_chain_parts_jerk_wrench_and_ext_torque(case, layout):
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import trajectory_batch, uniform_pm1
    chain, ref = _pair(case)
    N, n, L = 523, ref.n, ref.L
    q, dq, ddq, dddq = trajectory_batch(23, N, n, order=4)
    ext = 4.0 * uniform_pm1(31, (N, L, 6))
    dev, host = _io(torch, layout)
    tq, tdq, tddq, tdddq, text = dev(q), dev(dq), dev(ddq), dev(dddq), dev(ext)
    a, al, an = ref.dtwist(q, dq, ddq, parts=True)
    _close(host(chain.getDTwistLinearPart(tq, tddq, layout=layout)), al, "linear part")
    _close(host(chain.getDTwistNonLinearPart(tq, tdq, layout=layout)), an, "non-linear part")
    _close(host(chain.getDDTwist(tq, tdq, tddq, tdddq, layout=layout)), ref.ddtwist(q, dq, ddq, dddq), "jerk")
    jl, jn = ref.ddtwist_parts(q, dq, ddq, dddq)
    _close(host(chain.getDDTwistLinearPart(tq, tdddq, layout=layout)), jl, "jerk linear part")
    _close(host(chain.getDDTwistNonLinearPart(tq, tdq, tddq, layout=layout)), jn, "jerk non-linear part")
    tau, w = ref.joint_torque(q, dq, ddq, ext=ext, wrenches=True)
    _close(host(chain.getWrench(tq, tdq, tddq, text, layout=layout)), w, "wrenches with external loads")
    tau0, w0 = ref.joint_torque(q, dq, ddq, wrenches=True)
    _close(host(chain.getWrench(tq, tdq, tddq, layout=layout)), w0, "wrenches")
    _close(host(chain.getJointTorqueExt(tq, tdq, tddq, text, layout=layout)), tau, "tau ext")
    z = torch.zeros_like(text)
    _close(host(chain.getJointTorqueExt(tq, tdq, tddq, z, layout=layout)), tau0, "tau ext (zero loads)")
    if n <= 10:
        # the companion's torque (fixed frames folded into the bodies) and the wrench recursion over all links agree
        _close(host(chain.getJointTorque(tq, tdq, tddq, layout=layout)), tau0, "tau through the companion")


@pytest.mark.parametrize("case", ["ur10_long_permuted", "gen20_permuted"])
def test_long_chain_permuted_inputs_regressor_torque_inertia(case):
    """Input joints in any order on a chain longer than the kernels sweep: the reduced companion keeps the joints in chain order and
    carries the input index of each (primitives_impl.h:724-737)."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _pair(case)
    N, n, P = 640, ref.n, ref.P
    q, dq, ddq = trajectory_batch(41, N, n)
    Yr, tr = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    eq, edq, eddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))
    Y, tau = chain.getRegressor(eq, edq, eddq, layout="element", with_torque=True)
    _close(Y.cpu().numpy().transpose(2, 1, 0), Yr, "Y element")
    _close(tau.cpu().numpy().T, tr, "tau fused")
    Y, tau = chain.getRegressor(tq, tdq, tddq, with_torque=True)
    _close(Y.cpu().numpy().transpose(0, 2, 1), Yr, "Y per-sample")
    _close(tau.cpu().numpy(), tr, "tau")
    Ys = chain.getRegressor(tq, tdq, tddq, y_layout="stacked")
    _close(Ys.cpu().numpy().reshape(P, N, n).transpose(1, 2, 0), Yr, "Y stacked")
    _close(chain.getJointTorque(tq, tdq, tddq).cpu().numpy(), tr, "getJointTorque")
    _close(chain.getJointInertia(eq, layout="element").cpu().numpy().transpose(2, 1, 0), ref.joint_inertia(q), "M")


def generated_revolute_chain(nj, seed):
    """nj REVOLUTE joints, every one an input joint, every link with inertial data."""
    from rosdyn_amd.samples import uniform_pm1
    r = uniform_pm1(seed, (nj + 1, 16))
    links, joints = ["<link name='l0'/>"], []
    for i in range(nj):
        k = r[i]
        joints.append(
            "<joint name='j%d' type='revolute'><parent link='l%d'/><child link='l%d'/>"
            "<origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g %.17g'/><axis xyz='%.17g %.17g %.17g'/>"
            "<limit lower='-3' upper='3' effort='10' velocity='2'/></joint>"
            % (i, i, i + 1, 0.1 * k[0], 0.1 * k[1], 0.08 + 0.05 * k[2], k[3], k[4], k[5], k[6], k[7], 1.0 + 0.5 * k[8]))
        links.append(
            "<link name='l%d'><inertial><origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g 0'/><mass value='%.17g'/>"
            "<inertia ixx='%.17g' ixy='%.17g' ixz='%.17g' iyy='%.17g' iyz='%.17g' izz='%.17g'/></inertial></link>"
            % (i + 1, 0.05 * k[9], 0.05 * k[10], 0.05 * k[11], 0.3 * k[12], 0.3 * k[13], 1.2 + k[14],
               0.02, 0.002 * k[15], -0.001, 0.03, 0.0015, 0.025))
    return "<robot name='rev%d'>%s%s</robot>" % (nj, "".join(links), "".join(joints))


@pytest.mark.parametrize("case", ["rev14", "rev20", "rev32", "gen20_all", "gen32_all", "gen20_twelve_permuted"])
def test_long_chain_more_than_ten_input_joints_regressor_torque_inertia(case):
    """Round 6: regressor (+ fused torque) and joint inertia of chains with MORE than ten input joints -- the run-time-length kernels of
    rdyn_long_local.hip (rolled link and row loops, per-joint state in wave-private LDS) -- against the C oracle, all three regressor
    layouts, both input layouts; N = 200 leaves a ragged wave.  Round 5 answered RDYN_ERR_UNSUPPORTED here.  The reference's default
    build has no bound on the number of joints (rosdyn_core/CMakeLists.txt:12-16, primitives_impl.h:1295-1379)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    inputs = None
    if case.startswith("rev"):
        nj = int(case[3:])
        xml, base, tool = generated_revolute_chain(nj, 1000 + nj), "l0", "l%d" % nj
    elif case == "gen20_twelve_permuted":
        xml, base, tool = generated_long_chain(20, 2020), "l0", "l20"
        inputs = ["j13", "j0", "j9", "j4", "j16", "j1", "j7", "j19", "j3", "j10", "j6", "j12"]   # 12 of the 14 moving joints, out of chain order
    else:
        xml, base, tool, inputs = _cases()[case]
    chain, ref = Chain(xml, base, tool, GRAV), OracleChain(xml, base, tool, GRAV, input_joint_names=inputs)
    if inputs:
        assert chain.setInputJointsName(inputs)
    n, P = ref.n, ref.P
    assert n > 10 and chain.getActiveJointsNumber() == n
    N = 200
    q, dq, ddq = trajectory_batch(61, N, n)
    Yr, tr, Mr = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq), ref.joint_inertia(q)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    eq, edq, eddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))
    Y, tau = chain.getRegressor(eq, edq, eddq, layout="element", with_torque=True)
    _close(Y.cpu().numpy().transpose(2, 1, 0), Yr, "Y element")
    _close(tau.cpu().numpy().T, tr, "tau fused (element)")
    Y, tau = chain.getRegressor(tq, tdq, tddq, with_torque=True)
    _close(Y.cpu().numpy().transpose(0, 2, 1), Yr, "Y per-sample")
    _close(tau.cpu().numpy(), tr, "tau fused")
    Ys = chain.getRegressor(tq, tdq, tddq, y_layout="stacked")
    _close(Ys.cpu().numpy().reshape(P, N, n).transpose(1, 2, 0), Yr, "Y stacked")
    # Y pi = tau with the chain's own nominal parameters (the identity of SURVEY.md section 4)
    _close(np.einsum("snp,p->sn", Y.cpu().numpy().transpose(0, 2, 1), chain.getNominalParameters()), tr, "Y pi")
    _close(chain.getJointInertia(eq, layout="element").cpu().numpy().transpose(2, 1, 0), Mr, "M element")
    _close(chain.getJointInertia(tq).cpu().numpy().transpose(0, 2, 1), Mr, "M sample")
    # the joint torques of such chains: the wrench recursion of the run-time-length kernels (primitives_impl.h:1264-1272)
    _close(chain.getJointTorque(tq, tdq, tddq).cpu().numpy(), tr, "getJointTorque")
    _close(chain.getJointTorqueNonLinearPart(tq, tdq).cpu().numpy(), ref.joint_torque(q, dq, 0 * ddq), "non-linear part")
    # every getter of the samples through the one entry point (rdyn_evaluate_all serves long chains by the single-purpose launches)
    o = chain.evaluateAll(tq, tdq, tddq)
    _close(o["Y"].cpu().numpy().transpose(0, 2, 1), Yr, "evaluateAll Y")
    _close(o["M"].cpu().numpy().transpose(0, 2, 1), Mr, "evaluateAll M")
    _close(o["tau"].cpu().numpy(), tr, "evaluateAll tau")
    _close(o["T_links"].cpu().numpy().transpose(0, 1, 3, 2), ref.fk(q), "evaluateAll T")


def test_eleven_input_joints_normal_equations_and_r_factor():
    """rdyn_regressor_gram / rdyn_regressor_tsqr of a chain with ELEVEN input joints (110 columns + tau_meas = the 111 the Gram kernel holds): chunk images by
    rdyn_long_local.hip contracted by k_gram, several chunks with a ragged last one, accumulation; 12 input joints (121 columns) are
    refused before anything touches the device (the documented limit)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    from rosdyn_amd.samples import trajectory_batch
    xml = generated_revolute_chain(11, 1111)
    chain, ref = Chain(xml, "l0", "l11", GRAV), OracleChain(xml, "l0", "l11", GRAV)
    n, P, N = 11, 110, 3000
    q, dq, ddq = trajectory_batch(71, N, n)
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(2).normal(size=(N, n))
    A = ref.regressor(q, dq, ddq).reshape(N * n, P)
    Gr, cr, bbr = A.T @ A, A.T @ tau.reshape(-1), float(tau.reshape(-1) @ tau.reshape(-1))
    tq, tdq, tddq, tt = (torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))
    for chunk in (0, 1024):   # one chunk / three chunks with a ragged last one
        G, c, bb = chain.getRegressorGram(tq, tdq, tddq, tt, chunk_samples=chunk)
        assert np.linalg.norm(G.cpu().numpy() - Gr) <= 1e-10 * np.linalg.norm(Gr)
        assert np.linalg.norm(c.cpu().numpy() - cr) <= 1e-10 * np.linalg.norm(cr)
        assert abs(float(bb.cpu()[0]) - bbr) <= 1e-10 * bbr
    eq, edq, eddq, et = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq, tau))
    G, c, bb = chain.getRegressorGram(eq, edq, eddq, et, layout="element", chunk_samples=1000)
    assert np.linalg.norm(G.cpu().numpy() - Gr) <= 1e-10 * np.linalg.norm(Gr) and np.linalg.norm(c.cpu().numpy() - cr) <= 1e-10 * np.linalg.norm(cr)
    # the R factor of [A | tau] without the normal equations (rdyn_regressor_tsqr: chunk images -> rdyn_tsqr's kernels, 111 columns)
    M = np.column_stack([A, tau.reshape(-1)])
    full = M.T @ M
    R1 = chain.getRegressorTsqr(tq, tdq, tddq, tt).cpu().numpy()
    assert R1.shape == (P + 1, P + 1) and np.allclose(np.tril(R1, -1), 0.0)
    assert np.abs(R1.T @ R1 - full).max() <= 1e-10 * np.abs(full).max()
    R1e = chain.getRegressorTsqr(eq, edq, eddq, et, layout="element").cpu().numpy()
    assert np.abs(R1e.T @ R1e - full).max() <= 1e-10 * np.abs(full).max()
    twelve = Chain(generated_revolute_chain(12, 1212), "l0", "l12", GRAV)
    assert lib().rdyn_regressor_gram_workspace_bytes(twelve._h, 0) == 0 and lib().rdyn_regressor_tsqr_workspace_bytes(twelve._h) == 0


@pytest.mark.parametrize("case", ["ur10_long", "gen20_permuted"])
def test_long_chain_local_ik(case):
    """computeLocalIk on the companion + the constant frames behind the last input joint, against the oracle iterating the whole chain."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain, frame_distance
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import uniform_pm1
    xml, base, tool, inputs = _cases()[case]
    if case == "gen20_permuted":
        inputs = inputs[:6]          # six input joints: J'J can be definite
    chain, ref = Chain(xml, base, tool), OracleChain(xml, base, tool, input_joint_names=inputs)
    if inputs:
        assert chain.setInputJointsName(inputs)
    N = 800
    lo, hi = np.array(ref.spec.q_min), np.array(ref.spec.q_max)
    lo, hi = np.maximum(lo, -3.0), np.minimum(hi, 3.0)
    q_goal = np.clip(uniform_pm1(7, (N, ref.n)), lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo))
    seeds = np.clip(q_goal + 0.2 * uniform_pm1(8, (N, ref.n)), lo, hi)
    T = ref.fk(q_goal)[:, -1]
    Tt = torch.from_numpy(np.ascontiguousarray(T.transpose(0, 2, 1))).cuda()
    sol, st, it = chain.computeLocalIk(Tt, torch.from_numpy(np.ascontiguousarray(seeds)).cuda(), toll=1e-6, max_iterations=30)
    sol, st, it = sol.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    rsol, rst, rit = ref.local_ik(T, seeds, toll=1e-6, max_iter=30)
    conv = (rst == 1) & (rit <= 8)
    assert conv.mean() > 0.6, conv.mean()
    assert (st[conv] == 1).all()
    assert np.array_equal(it[conv], rit[conv])
    assert np.abs(sol[conv] - rsol[conv]).max() < 1e-9
    Ts = ref.fk(sol[conv])[:, -1]
    worst = max(np.linalg.norm(frame_distance(a, b)) for a, b in zip(T[conv][:200], Ts[:200]))
    assert worst < 1e-6
