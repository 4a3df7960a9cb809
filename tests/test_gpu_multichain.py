"""GPU test of the mixed-chain batch (BASELINE.json configs[4]): distinct perturbed 6-/7-DOF chains, one launch per
joint-count group, checked item by item against the CPU oracle on a sample prefix and against the single-chain path."""
import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)


def test_mixed_chains_match_oracle_and_single_chain_path():
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.samples import trajectory_batch
    from rosdyn_amd.urdf_gen import mixed_chain_set
    specs = mixed_chain_set(FIXTURES, n_chains=12)
    assert len({s[0] for s in specs}) == 12                     # all distinct
    S = [700, 256, 1, 1000, 513, 64, 300, 300, 300, 300, 2, 999]  # ragged batch sizes
    items, refs, ins = [], [], []
    for i, (xml, base, tool) in enumerate(specs):
        chain = Chain(xml, base, tool, GRAV)
        q, dq, ddq = trajectory_batch(1000 + i, S[i], chain.getActiveJointsNumber())
        tq, tdq, tddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq))
        items.append((chain, tq, tdq, tddq))
        refs.append(OracleChain(xml, base, tool, GRAV))
        ins.append((q, dq, ddq))
    plan = MultiChainRegressor(items)
    Y, tau = plan.run()
    torch.cuda.synchronize()
    for i, (chain, tq, tdq, tddq) in enumerate(items):
        k = min(S[i], 64)
        q, dq, ddq = (x[:k] for x in ins[i])
        Yr, tr = refs[i].regressor(q, dq, ddq), refs[i].joint_torque(q, dq, ddq)
        Yg = Y[i].cpu().numpy().transpose(2, 1, 0)[:k]
        tg = tau[i].cpu().numpy().T[:k]
        assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())
        assert np.abs(tg - tr).max() <= 1e-11 * max(1.0, np.abs(tr).max())
        Y1, t1 = chain.getRegressor(tq, tdq, tddq, layout="element", with_torque=True)
        assert torch.equal(Y1, Y[i]) and torch.equal(t1, tau[i])    # same kernel body, bit identical


@pytest.mark.parametrize("y_layout", ["per_sample", "stacked"])
def test_mixed_chains_in_the_row_contiguous_layouts(y_layout):
    """The drop-in per-sample images and the stacked matrices from a mixed-chain plan (k_image_sweep_multi: LDS-staged, whole-line
    stores): bit-identical to the single-chain calls item by item, ragged batch sizes (empty waves of short items leave at once),
    nothing written outside an item's own buffer."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.samples import trajectory_batch
    from rosdyn_amd.urdf_gen import mixed_chain_set
    specs = mixed_chain_set(FIXTURES, n_chains=10)
    S = [700, 256, 1, 1000, 513, 64, 63, 65, 2, 999]
    items = []
    for i, (xml, base, tool) in enumerate(specs):
        chain = Chain(xml, base, tool, GRAV)
        q, dq, ddq = trajectory_batch(2000 + i, S[i], chain.getActiveJointsNumber())
        items.append((chain,) + tuple(torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq)))
    plan = MultiChainRegressor(items, y_layout=y_layout)
    for Y in plan.Y:
        Y.fill_(float("nan"))
    Y, tau = plan.run()
    torch.cuda.synchronize()
    for i, (chain, tq, tdq, tddq) in enumerate(items):
        Y1, t1 = chain.getRegressor(tq, tdq, tddq, layout="element", y_layout=y_layout, with_torque=True)
        assert not torch.isnan(Y[i]).any()
        assert torch.equal(Y1.reshape(Y[i].shape), Y[i]) and torch.equal(t1, tau[i])
