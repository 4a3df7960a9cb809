"""rdyn_evaluate_all (include/rdyn.h): every getter of a sample in ONE launch -- the seven sweeps side by side, the same device code as
the single-purpose kernels: bit-identical to them, and checked against the oracle.  (No reference counterpart: the reference caches
what a call computed on the way, primitives_impl.h:886, 985, 1088.)"""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.1, -0.2, -9.7)


@pytest.mark.parametrize("urdf,base,tool,inputs", [("ur10_like.urdf", "base_link", "tool0", None), ("mixed_joints.urdf", "world", "tip", None),
                                                   ("panda_like.urdf", "link0", "hand", ["joint5", "joint2", "joint7", "joint1"]),
                                                   ("ur10_public_long.urdf", "base_link", "tcp", None)])
@pytest.mark.parametrize("layout", ["sample", "element"])
@pytest.mark.parametrize("N", [1, 64, 4097])
def test_evaluate_all_equals_the_single_purpose_getters(urdf, base, tool, inputs, layout, N):
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV, input_joint_names=inputs)
    if inputs:
        assert chain.setInputJointsName(inputs)
    n = ref.n
    q, dq, ddq = trajectory_batch(N + 7, N, n)
    if layout == "element":
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    tq, tdq, tddq = dev(q), dev(dq), dev(ddq)
    o = chain.evaluateAll(tq, tdq, tddq, layout=layout)
    torch.cuda.synchronize()
    assert torch.equal(o["T_links"], chain.getTransformations(tq, layout=layout))
    assert torch.equal(o["J"], chain.getJacobian(tq, layout=layout))
    assert torch.equal(o["twists"], chain.getTwist(tq, tdq, layout=layout))
    assert torch.equal(o["dtwists"], chain.getDTwist(tq, tdq, tddq, layout=layout))
    assert torch.equal(o["tau"], chain.getJointTorque(tq, tdq, tddq, layout=layout))
    assert torch.equal(o["tau_nonlinear"], chain.getJointTorqueNonLinearPart(tq, tdq, layout=layout))
    assert torch.equal(o["M"], chain.getJointInertia(tq, layout=layout))
    host = (lambda t: np.moveaxis(t.cpu().numpy(), -1, 0)) if layout == "element" else (lambda t: t.cpu().numpy())

    def close(a, b, what):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max()), what
    Yr = ref.regressor(q, dq, ddq)
    close(host(o["Y"]).transpose(0, 2, 1), Yr, "Y")
    close(host(o["tau"]), ref.joint_torque(q, dq, ddq), "tau")
    close(host(o["T_links"]).transpose(0, 1, 3, 2), ref.fk(q), "T")
    close(host(o["M"]).transpose(0, 2, 1), ref.joint_inertia(q), "M")
    close(host(o["J"]).transpose(0, 2, 1), ref.jacobian(q), "J")
