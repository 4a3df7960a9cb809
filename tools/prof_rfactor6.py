#!/usr/bin/env python3
"""rocprofv3 target: the R factor of [Y | tau] at config-2 size (6 joints, N = 1e6), six calls -- the fixed cost of the route
(k_cholqr_precond, k_cholqr_factor, the launches that leave at once) next to pass B.  tools/gpu_profile.sh script <tag> tools/prof_rfactor6.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain  # noqa: E402

N, n = 1000000, 6
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
tau = chain.getJointTorque(q, dq, ddq, layout="element")
for _ in range(6):
    chain.getRegressorTsqr(q, dq, ddq, tau, layout="element")
torch.cuda.synchronize()
