#!/usr/bin/env python3
"""R factor of [Y | tau] for 9 and 10 input joints (91 / 101 columns): chunk images + rdyn_tsqr's kernels; time per call at N = 1e6."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain                 # noqa: E402
from tools.probe import timeit               # noqa: E402
from test_gpu_longchain import _chain_xml    # noqa: E402

N = 1000000
for nj in (9, 10):
    chain = Chain(_chain_xml(nj, 300 + nj), "l0", "l%d" % nj, (0.3, -0.4, -9.7))
    q, dq, ddq = (torch.rand((N, nj), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    tau = chain.getJointTorque(q, dq, ddq)
    t = timeit(lambda: chain.getRegressorTsqr(q, dq, ddq, tau), reps=3, warm=1)
    print("%2d input joints (%3d columns): R factor of [Y | tau], N = 1e6   %9.1f us" % (nj, 10 * nj + 1, t * 1e6), flush=True)
