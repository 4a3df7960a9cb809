#!/usr/bin/env python3
"""rocprofv3 workload: config-2 chain, fused regressor_gram at two chunk sizes + generic gram."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain      # noqa: E402
from rosdyn_amd.gram import gram  # noqa: E402

dev = torch.device("cuda:0")
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, P, N = 6, 60, 1000000
q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(4))
for ch in (131072, 1000000):
    for _ in range(3):
        chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=ch)
Y = chain.getRegressor(q, dq, ddq, layout="element")
for _ in range(3):
    gram(Y.reshape(P, n * N), tm.reshape(n * N))
torch.cuda.synchronize()
