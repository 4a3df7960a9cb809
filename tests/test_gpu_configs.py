"""GPU tests of the BASELINE.json configurations at their stated sizes and of the launch-splitting / 64-bit addressing paths
(VERDICT r1 item 1):
  (a) configs[4] as written: 256 distinct 6-/7-DOF chains x 4 096 samples through rdyn_multi_plan_regressor;
  (b) the > 4 GB outputs: row-pair kernel split into two launches (rdyn_api.cpp run_local) and the LDS-staged image kernel with
      64-bit wave bases, both at N = 1.5e6 in the per-sample layout;
  (c) configs[3]'s single-rank share end to end on the real HIP path: two processes on ONE GPU, each the fused regressor -> Gram of
      its shard, one all-reduce of the packed normal equations, against the one-rank Gram of the concatenated batch.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import FIXTURES, ROOT

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)


def test_config5_256_chains_x_4096_samples():
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    specs = mixed_chain_set(FIXTURES, n_chains=256)
    assert len({s[0] for s in specs}) == 256                    # 256 DISTINCT chains
    S = 4096
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0005)
    items, chains = [], []
    for xml, base, tool in specs:
        chain = Chain(xml, base, tool, GRAV)
        n = chain.getActiveJointsNumber()
        q, dq, ddq = (torch.rand((n, S), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3))
        items.append((chain, q, dq, ddq))
        chains.append(chain)
    assert {c.getJointsNumber() for c in chains} == {6, 7}      # two joint-count groups -> two launches
    plan = MultiChainRegressor(items)
    Y, tau = plan.run()
    torch.cuda.synchronize()
    total = 0
    for i, (chain, q, dq, ddq) in enumerate(items):
        P = 10 * chain.getJointsNumber()
        # identity on ALL 4 096 evaluations of the item: Y pi = tau (getRegressor x getNominalParameters == getJointTorque)
        pi = torch.from_numpy(chain.getNominalParameters()).cuda()
        res = torch.einsum("pjs,p->js", Y[i], pi) - tau[i]
        assert float(res.abs().max()) <= 1e-10 * max(1.0, float(tau[i].abs().max())), i
        assert Y[i].shape == (P, chain.getActiveJointsNumber(), S)
        total += S
        # oracle parity on a 64-sample prefix of EVERY item
        k = 64
        ref = OracleChain(specs[i][0], specs[i][1], specs[i][2], GRAV)
        qh, dqh, ddqh = (np.ascontiguousarray(x[:, :k].cpu().numpy().T) for x in (q, dq, ddq))
        Yr, tr = ref.regressor(qh, dqh, ddqh), ref.joint_torque(qh, dqh, ddqh)
        Yg = Y[i][:, :, :k].cpu().numpy().transpose(2, 1, 0)
        tg = tau[i][:, :k].cpu().numpy().T
        assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max()), i
        assert np.abs(tg - tr).max() <= 1e-11 * max(1.0, np.abs(tr).max()), i
    assert total == 1048576


@pytest.mark.parametrize("permuted", [True, False], ids=["rowpair_two_launches", "image_kernel_64bit_bases"])
def test_per_sample_layout_beyond_4GB(permuted):
    """N = 1.5e6, per-sample images of 2 880 B: 4.32 GB of regressor.  Input joints in another order -> the row-pair kernel, which
    addresses Y with 32-bit lane offsets and is therefore launched twice (split at sample 1 397 888); chain order -> k_image_sweep
    (64-bit wave bases).  Checked against the element-major kernel on the whole batch and against the oracle on samples that
    straddle the split and the 4 GB mark."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    urdf = os.path.join(FIXTURES, "ur10_like.urdf")
    names = None
    chain = Chain(urdf, "base_link", "wrist_3_link", GRAV)
    if permuted:
        names = list(reversed(chain.getActiveJointsName()))
        assert chain.setInputJointsName(names)
    n, P, N = 6, 60, 1500000
    gen = torch.Generator(device="cuda").manual_seed(77)
    q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3))
    Yp = torch.full((N, P, n), float("nan"), dtype=torch.float64, device="cuda")
    Yp, tau_p = chain.getRegressor(q, dq, ddq, y_layout="per_sample", out=Yp, with_torque=True)
    assert Yp.numel() * 8 > 2 ** 32
    qe, dqe, ddqe = (x.t().contiguous() for x in (q, dq, ddq))
    Ye, tau_e = chain.getRegressor(qe, dqe, ddqe, layout="element", with_torque=True)     # (P, n, N)
    torch.cuda.synchronize()
    split = (0xF0000000 // (n * P * 8)) & ~255
    assert 0 < split < N
    worst = 0.0
    for s0 in range(0, N, 250000):                                # compare in slabs (memory)
        a = Yp[s0:s0 + 250000]                                    # (k, P, n)
        b = Ye[:, :, s0:s0 + 250000].permute(2, 0, 1)
        assert not torch.isnan(a).any()
        worst = max(worst, float((a - b).abs().max()))
    assert worst <= 1e-12, worst
    assert float((tau_p - tau_e.t()).abs().max()) <= 1e-12
    ref = OracleChain(urdf, "base_link", "wrist_3_link", GRAV, names)
    for lo in (0, split - 32, (2 ** 32) // (n * P * 8) - 32, N - 64):
        sl = slice(lo, lo + 64)
        qh, dqh, ddqh = (x[sl].cpu().numpy() for x in (q, dq, ddq))
        Yr = ref.regressor(qh, dqh, ddqh)                          # (k, n, P)
        Yg = Yp[sl].cpu().numpy().transpose(0, 2, 1)
        assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max()), lo


def test_config4_share_two_ranks_on_one_gpu(tmp_path):
    """configs[3] end to end on the real HIP path, world size 2 on device 0: every rank runs rdyn_regressor_gram on its shard, packs
    [G | c | bb | count], ONE all-reduce (gloo: two ranks cannot share one device under RCCL), and rank 0 compares with the one-rank
    Gram of the concatenated batch."""
    script = os.path.join(ROOT, "tests", "_world2_gram.py")
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", script, str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    d = json.loads(out.read_text())
    assert d["world"] == 2 and d["count"] == d["n_total"]
    assert d["rel_G"] <= 1e-12 and d["rel_c"] <= 1e-12 and d["rel_bb"] <= 1e-12, d
    assert d["param_err"] <= 1e-6, d


def test_config3_share_two_ranks_on_one_gpu(tmp_path):
    """The R-factor exchange of SURVEY 8(e) end to end on the real HIP path, world size 2 on device 0: every rank the robust factor of
    [Y | 7 friction columns | tau] of its shard (7-joint arm with fixed flange / hand frames), ONE all-gather, the fold of the stack
    on every rank -- equal to the factor of the whole batch from one call and to the normal equations, the same bits on both ranks."""
    script = os.path.join(ROOT, "tests", "_world2_tsqr.py")
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29578", script, str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    d = json.loads(out.read_text())
    assert d["world"] == 2 and d["n1"] == 105 and d["upper"] and d["same_bits_on_all_ranks"], d
    assert d["rel_gram"] <= 1e-11 and d["rel_one_call"] <= 1e-11 and d["sv_err"] <= 1e-9, d
