"""GPU parity of the split / jerk sweeps and the external-wrench torque vs the C oracle."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)
TOL = 1e-11


def _close(a, b, what):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(np.asarray(a) - b).max())
    assert err <= TOL * scale, "%s: %.3e > %.1e * %.3g" % (what, err, TOL, scale)


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "tool0"), ("mixed_joints.urdf", "world", "tip"),
                                            ("panda_like.urdf", "link0", "hand")])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_parts_jerk_and_ext_wrench(urdf, base, tool, layout):
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch, uniform_pm1
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    N, n, L = 1500, ref.n, ref.L
    q, dq, ddq, dddq = trajectory_batch(321, N, n, order=4)
    ext = 5.0 * uniform_pm1(654, (N, L, 6))
    if layout == "element":
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()
        host = lambda t: np.moveaxis(t.cpu().numpy(), -1, 0)
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        host = lambda t: t.cpu().numpy()
    tq, tdq, tddq, tdddq, text = dev(q), dev(dq), dev(ddq), dev(dddq), dev(ext)
    a, al, an = ref.dtwist(q, dq, ddq, parts=True)
    _close(host(chain.getDTwistLinearPart(tq, tddq, layout=layout)), al, "linear part")
    _close(host(chain.getDTwistNonLinearPart(tq, tdq, layout=layout)), an, "non-linear part")
    _close(host(chain.getDDTwist(tq, tdq, tddq, tdddq, layout=layout)), ref.ddtwist(q, dq, ddq, dddq), "jerk")
    _close(host(chain.getJointTorqueExt(tq, tdq, tddq, text, layout=layout)), ref.joint_torque(q, dq, ddq, ext=ext), "tau ext")
    # zero external wrenches == plain getJointTorque
    z = torch.zeros_like(text)
    assert torch.equal(chain.getJointTorqueExt(tq, tdq, tddq, z, layout=layout), chain.getJointTorque(tq, tdq, tddq, layout=layout))


@pytest.mark.parametrize("urdf,base,tool,inputs", [
    ("ur10_like.urdf", "base_link", "tool0", None),
    ("mixed_joints.urdf", "world", "tip", None),
    # permuted subset of the moveable joints: the reference fills the first joints.size() input columns
    ("panda_like.urdf", "link0", "hand", ["joint5", "joint2", "joint7", "joint1"])])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_by_link_getters(urdf, base, tool, inputs, layout):
    """getJacobianLink / getTransformationLink / getTwistLink (primitives_impl.h:914-925, 951-979, 1016-1027)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV, input_joint_names=inputs)
    if inputs:
        names = [j for j in inputs if j in chain.getMoveableJointNames()]
        assert names == inputs and chain.setInputJointsName(inputs)
    N, n = 777, ref.n
    q, dq, _ = trajectory_batch(99, N, n)
    if layout == "element":
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()
        host = lambda t: np.moveaxis(t.cpu().numpy(), -1, 0)
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        host = lambda t: t.cpu().numpy()
    tq, tdq = dev(q), dev(dq)
    T, tw = ref.fk(q), ref.twist(q, dq)
    links = chain.getLinksName()
    for i, name in enumerate(links):
        J = host(chain.getJacobianLink(tq, name, layout=layout)).transpose(0, 2, 1)
        _close(J, ref.jacobian_link(q, i), "J link %s" % name)
        _close(host(chain.getTransformationLink(tq, name, layout=layout)).transpose(0, 2, 1), T[:, i], "T link %s" % name)
        _close(host(chain.getTwistLink(tq, tdq, name, layout=layout)), tw[:, i], "twist link %s" % name)
    # the tool link's Jacobian is getJacobian, bit for bit
    assert torch.equal(chain.getJacobianLink(tq, links[-1], layout=layout), chain.getJacobian(tq, layout=layout))
    with pytest.raises(ValueError, match="is not member of the chain"):
        chain.getJacobianLink(tq, "no_such_link", layout=layout)


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "tool0"), ("mixed_joints.urdf", "world", "tip"),
                                            ("panda_like.urdf", "link0", "hand")])
@pytest.mark.parametrize("layout", ["sample", "element"])
def test_jerk_split_and_link_wrenches(urdf, base, tool, layout):
    """getDDTwistLinearPart / getDDTwistNonLinearPart (primitives_impl.h:1126-1183) and getWrench (:1225-1262, forward-only
    accumulation on the GPU vs the reference's tool -> base recursion in the oracle)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch, uniform_pm1
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    N, n, L = 1100, ref.n, ref.L
    q, dq, ddq, dddq = trajectory_batch(77, N, n, order=4)
    ext = 4.0 * uniform_pm1(88, (N, L, 6))
    if layout == "element":
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()
        host = lambda t: np.moveaxis(t.cpu().numpy(), -1, 0)
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        host = lambda t: t.cpu().numpy()
    tq, tdq, tddq, tdddq, text = dev(q), dev(dq), dev(ddq), dev(dddq), dev(ext)
    jl, jn = ref.ddtwist_parts(q, dq, ddq, dddq)
    _close(host(chain.getDDTwistLinearPart(tq, tdddq, layout=layout)), jl, "jerk linear part")
    _close(host(chain.getDDTwistNonLinearPart(tq, tdq, tddq, layout=layout)), jn, "jerk non-linear part")
    _close(jl + jn, ref.ddtwist(q, dq, ddq, dddq), "oracle: parts sum to the jerk")
    tau, w = ref.joint_torque(q, dq, ddq, ext=ext, wrenches=True)
    _close(host(chain.getWrench(tq, tdq, tddq, text, layout=layout)), w, "wrenches with external loads")
    _, w0 = ref.joint_torque(q, dq, ddq, wrenches=True)
    _close(host(chain.getWrench(tq, tdq, tddq, layout=layout)), w0, "wrenches")
