"""GPU parity on generated chains at the ends of the supported range (1, 2, 3, 9 and 10 chain joints): exercises every
kernel instantiation boundary (NJ = 10 regressor spills to scratch, 7 Gram column blocks) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GRAV = (0.3, -0.4, -9.7)


def _chain_xml(nj, seed):
    from rosdyn_amd.samples import uniform_pm1
    r = uniform_pm1(seed, (nj + 1, 16))
    links = ["<link name='l0'/>"]
    joints = []
    for i in range(nj):
        k = r[i]
        typ = "prismatic" if (i % 4 == 3) else "revolute"
        joints.append(
            "<joint name='j%d' type='%s'><parent link='l%d'/><child link='l%d'/>"
            "<origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g %.17g'/><axis xyz='%.17g %.17g %.17g'/>"
            "<limit lower='-3' upper='3' effort='10' velocity='2'/></joint>"
            % (i, typ, i, i + 1, 0.2 * k[0], 0.2 * k[1], 0.15 + 0.1 * k[2], k[3], k[4], k[5], k[6], k[7], 1.0 + 0.5 * k[8]))
        links.append(
            "<link name='l%d'><inertial><origin xyz='%.17g %.17g %.17g' rpy='%.17g %.17g 0'/><mass value='%.17g'/>"
            "<inertia ixx='%.17g' ixy='%.17g' ixz='%.17g' iyy='%.17g' iyz='%.17g' izz='%.17g'/></inertial></link>"
            % (i + 1, 0.05 * k[9], 0.05 * k[10], 0.05 * k[11], 0.3 * k[12], 0.3 * k[13], 1.5 + k[14],
               0.02, 0.002 * k[15], -0.001, 0.03, 0.0015, 0.025))
    return "<robot name='gen%d'>%s%s</robot>" % (nj, "".join(links), "".join(joints))


@pytest.mark.parametrize("nj", [1, 2, 3, 9, 10])
def test_generated_chain_parity(nj):
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    xml = _chain_xml(nj, 100 + nj)
    chain, ref = Chain(xml, "l0", "l%d" % nj, GRAV), OracleChain(xml, "l0", "l%d" % nj, GRAV)
    N, n, P = 700, ref.n, ref.P
    q, dq, ddq = trajectory_batch(5 + nj, N, n)

    def close(a, b, what):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max()), what

    Yr, tr = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq)
    for layout in ("sample", "element"):
        if layout == "element":
            tq, tdq, tddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))
            Y, tau = chain.getRegressor(tq, tdq, tddq, layout="element", with_torque=True)
            close(Y.cpu().numpy().transpose(2, 1, 0), Yr, "Y element")
            close(tau.cpu().numpy().T, tr, "tau fused")
            close(chain.getJointTorque(tq, tdq, tddq, layout="element").cpu().numpy().T, tr, "tau")
            close(chain.getJointInertia(tq, layout="element").cpu().numpy().transpose(2, 1, 0), ref.joint_inertia(q), "M")
            close(np.moveaxis(chain.getTransformations(tq, layout="element").cpu().numpy(), -1, 0).transpose(0, 1, 3, 2), ref.fk(q), "T")
            G, c, bb = chain.getRegressorGram(tq, tdq, tddq, tau, layout="element", chunk_samples=256)
            A = Yr.transpose(1, 0, 2).reshape(n * N, P)
            Gr = A.T @ A
            assert np.linalg.norm(G.cpu().numpy() - Gr) <= 1e-10 * np.linalg.norm(Gr)
        else:
            tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
            Y, tau = chain.getRegressor(tq, tdq, tddq, with_torque=True)              # per-sample (row-pair kernel)
            close(Y.cpu().numpy().transpose(0, 2, 1), Yr, "Y per-sample")
            close(tau.cpu().numpy(), tr, "tau rowpair")
            Ys = chain.getRegressor(tq, tdq, tddq, y_layout="stacked")
            close(Ys.cpu().numpy().reshape(P, N, n).transpose(1, 2, 0), Yr, "Y stacked")
