"""The batched entry points allocate, copy and synchronise nothing after a chain's first use on a device: they can be
captured into a HIP graph and replayed (include/rdyn.h, 'Memory')."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu


def test_regressor_and_gram_replay_from_a_graph():
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0", (0, 0, -9.806))
    n, P, N = 6, 70, 20000
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    Y = torch.empty((P, n, N), dtype=torch.float64, device="cuda")
    tau = torch.empty((n, N), dtype=torch.float64, device="cuda")
    out = (torch.empty((P, P), dtype=torch.float64, device="cuda"), torch.empty((P,), dtype=torch.float64, device="cuda"),
           torch.empty((1,), dtype=torch.float64, device="cuda"))
    from rosdyn_amd._lib import lib
    ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 8192),), dtype=torch.uint8, device="cuda")
    # first use uploads the chain constants (the only allocation/copy the library ever makes)
    chain.getRegressor(q, dq, ddq, layout="element", out=Y, tau_out=tau)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            chain.getRegressor(q, dq, ddq, layout="element", out=Y, tau_out=tau)
            chain.getRegressorGram(q, dq, ddq, tau, layout="element", chunk_samples=8192, out=out, workspace=ws)
    q.uniform_(-1, 1)                       # new inputs, same buffers
    Y.zero_()
    g.replay()
    torch.cuda.synchronize()
    Y2, tau2 = chain.getRegressor(q, dq, ddq, layout="element", with_torque=True)
    G2, c2, bb2 = chain.getRegressorGram(q, dq, ddq, tau2, layout="element", chunk_samples=8192)
    assert torch.equal(Y, Y2) and torch.equal(tau, tau2)
    assert torch.equal(out[0], G2) and torch.equal(out[1], c2) and torch.equal(out[2], bb2)


def test_pipelined_gram_ik_and_wrench_replay_from_a_graph():
    """The fused (LDS / pipelined) Gram path, the staged IK launches and the wrench sweep are capturable too."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
    n, L, P, N = 6, 7, 60, 30000
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    tau = chain.getJointTorque(q, dq, ddq, layout="element")           # first use: uploads the chain constants
    out = (torch.empty((P, P), dtype=torch.float64, device="cuda"), torch.empty((P,), dtype=torch.float64, device="cuda"),
           torch.empty((1,), dtype=torch.float64, device="cuda"))
    ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device="cuda")
    chain.getRegressorGram(q, dq, ddq, tau, layout="element", out=out, workspace=ws)   # sets the LDS opt-in attribute once
    T = chain.getTransformation(q, layout="element")
    seeds = q + 0.2 * (torch.rand_like(q) * 2 - 1)
    sol = torch.empty_like(q)
    w = torch.empty((L, 6, N), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    holder = {}
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            chain.getRegressorGram(q, dq, ddq, tau, layout="element", out=out, workspace=ws)
            holder["ik"] = chain.computeLocalIk(T, seeds, toll=1e-8, max_iterations=25, layout="element", out=sol)
            chain.getWrench(q, dq, ddq, layout="element", out=w)
    dq.uniform_(-1, 1)
    seeds.copy_(q + 0.1 * (torch.rand_like(q) * 2 - 1))
    g.replay()
    torch.cuda.synchronize()
    G2, c2, bb2 = chain.getRegressorGram(q, dq, ddq, tau, layout="element")
    assert torch.equal(out[0], G2) and torch.equal(out[1], c2) and torch.equal(out[2], bb2)
    sol2, st2, it2 = chain.computeLocalIk(T, seeds, toll=1e-8, max_iterations=25, layout="element")
    assert torch.equal(sol, sol2) and torch.equal(holder["ik"][1], st2) and torch.equal(holder["ik"][2], it2)
    assert torch.equal(w, chain.getWrench(q, dq, ddq, layout="element"))


def test_robust_factor_decides_on_the_device_inside_a_replayed_graph():
    """rdyn_regressor_tsqr above 4 096 samples (preconditioned CholeskyQR): the second round and the stand-by are launches that are
    always queued and leave at once unless a device flag says otherwise -- so ONE captured graph serves a batch the first round
    is accepted for and, replayed on new data in the same buffers, a batch that needs the second round (the tiles the preconditioner
    is built from are static poses).  Both factors must be numpy's."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd._lib import lib
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    grav = (0.0, 0.0, -9.806)
    chain, ref = Chain(path, "base_link", "wrist_3_link", grav), OracleChain(path, "base_link", "wrist_3_link", grav)
    n, P, N = 6, 60, 66000
    qa, dqa, ddqa = trajectory_batch(1, N, n)
    taua = ref.joint_torque(qa, dqa, ddqa) + 1e-3 * np.random.default_rng(1).normal(size=(N, n))
    qb, dqb, ddqb = trajectory_batch(2, N, n)
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024)
    stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    sub = (np.arange(N) // 16) % stride == 0
    dqb[sub] = 0.0
    ddqb[sub] = 0.0
    taub = ref.joint_torque(qb, dqb, ddqb) + 1e-3 * np.random.default_rng(2).normal(size=(N, n))
    bufs = [torch.from_numpy(x).cuda() for x in (qa, dqa, ddqa, taua)]
    ws = torch.empty((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h),), dtype=torch.uint8, device="cuda")
    chain.getRegressorTsqr(*bufs, workspace=ws)      # first use: chain constants, LDS opt-in attributes
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    holder = {}
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            holder["R"] = chain.getRegressorTsqr(*bufs, workspace=ws)
    for (q, dq, ddq, tau), stage in (((qa, dqa, ddqa, taua), 0), ((qb, dqb, ddqb, taub), 1), ((qa, dqa, ddqa, taua), 0)):
        for dst, src in zip(bufs, (q, dq, ddq, tau)):
            dst.copy_(torch.from_numpy(src))
        g.replay()
        torch.cuda.synchronize()
        R = holder["R"].cpu().numpy()
        assert chain.lastTsqrReport(N, ws)["stage"] == stage
        M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), tau.reshape(-1)])
        G = M.T @ M
        assert np.allclose(np.tril(R, -1), 0.0) and np.abs(R.T @ R - G).max() <= 1e-13 * np.abs(G).max()
        s_ref, s_gpu = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False), np.linalg.svd(R, compute_uv=False)
        keep = s_ref > 1e-9 * s_ref[0]
        assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-10
