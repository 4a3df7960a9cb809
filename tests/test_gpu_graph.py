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
