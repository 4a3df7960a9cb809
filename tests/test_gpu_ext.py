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
