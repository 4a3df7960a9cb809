#!/usr/bin/env python3
"""rocprofv3 workload: the default regressor->Gram kernel (wave-pair kernel since round 2) on config 2 and config 3, a few launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain      # noqa: E402

chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, N = 6, 1000000
q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
for _ in range(4):
    chain.getRegressorGram(q, dq, ddq, tm, layout="element")
torch.cuda.synchronize()

chain3 = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", "link7", (0, 0, -9.806))
N3 = 4000000
q3, dq3, ddq3, tm3 = (torch.rand((N3, 7), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
for _ in range(3):
    chain3.getRegressorGram(q3, dq3, ddq3, tm3)
torch.cuda.synchronize()
