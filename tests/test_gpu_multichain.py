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


@pytest.mark.parametrize("y_layout", ["per_sample", "stacked"])
def test_plan_with_fixed_joint_patterns(y_layout):
    """Round 3: the plan kernels are compiled per fixed-joint pattern (k_image_sweep_multi<NJ, FIX>).  One plan with six different
    patterns -- all joints moving, a fixed tool frame, the public-topology UR10 (fixed head joint; + flange; + flange and tool0), a
    Panda with its hand, a chain whose last moving joint is not an input joint -- in the row-contiguous layouts: every item against
    the oracle, and bit for bit against the single-chain call (same kernel body)."""
    torch = pytest.importorskip("torch")
    import os
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.samples import trajectory_batch
    cases = [("ur10_like.urdf", "base_link", "wrist_3_link", None), ("ur10_like.urdf", "base_link", "tool0", None),
             ("ur10_public.urdf", "base_link", "wrist_3_link", None), ("ur10_public.urdf", "base_link", "flange", None),
             ("ur10_public.urdf", "base_link", "tool0", None), ("panda_like.urdf", "link0", "hand", None),
             ("ur10_public.urdf", "base_link", "tool0", ["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint", "wrist_2_joint"]),
             ("ur10_public.urdf", "base_link", "tool0", None)]          # the same pattern twice: one group, two items
    S = [300, 129, 64, 1000, 513, 200, 77, 65]
    items, refs, ins = [], [], []
    for i, (urdf, base, tool, names) in enumerate(cases):
        path = os.path.join(FIXTURES, urdf)
        chain = Chain(path, base, tool, GRAV)
        if names:
            assert chain.setInputJointsName(names)
        refs.append(OracleChain(path, base, tool, GRAV, input_joint_names=names))
        q, dq, ddq = trajectory_batch(3000 + i, S[i], chain.getActiveJointsNumber())
        ins.append((q, dq, ddq))
        items.append((chain,) + tuple(torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq)))
    plan = MultiChainRegressor(items, y_layout=y_layout)
    for Y in plan.Y:
        Y.fill_(float("nan"))
    Y, tau = plan.run()
    torch.cuda.synchronize()
    for i, (chain, tq, tdq, tddq) in enumerate(items):
        n, P, N = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber(), S[i]
        Y1, t1 = chain.getRegressor(tq, tdq, tddq, layout="element", y_layout=y_layout, with_torque=True)
        assert not torch.isnan(Y[i]).any()
        assert torch.equal(Y1.reshape(Y[i].shape), Y[i]) and torch.equal(t1, tau[i])
        Yr = refs[i].regressor(*ins[i])                                  # (N, n, P)
        h = Y[i].cpu().numpy()
        Yg = h.transpose(0, 2, 1) if y_layout == "per_sample" else h.reshape(P, N, n).transpose(1, 2, 0)
        assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())
