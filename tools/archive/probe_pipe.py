#!/usr/bin/env python3
"""GPU probe: software-pipelined LDS Gram kernel (default) vs the two-phase one (RDYN_GRAM_PATH=lds0), same box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain          # noqa: E402
from rosdyn_amd._lib import lib       # noqa: E402
from tools.probe import timeit        # noqa: E402

for name, urdf, base, tool, N in (("cfg2 n=6 P=60", "ur10_like.urdf", "base_link", "wrist_3_link", 1000000),
                                  ("n=6 P=70 (tool0)", "ur10_like.urdf", "base_link", "tool0", 1000000),
                                  ("cfg3 n=7 P=70", "panda_like.urdf", "link0", "link7", 4000000),
                                  ("n=7 P=90 (hand)", "panda_like.urdf", "link0", "hand", 1000000)):
    chain = Chain(os.path.join(ROOT, "tests/fixtures", urdf), base, tool, (0, 0, -9.806))
    n = chain.getActiveJointsNumber()
    q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    ref = None
    for rep in range(2):
        for label, env in (("pipelined", {}), ("two-phase (lds0)", {"RDYN_GRAM_PATH": "lds0"})):
            for k, v in env.items():
                os.environ[k] = v
            ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device="cuda")
            out = chain.getRegressorGram(q, dq, ddq, tm, layout="element", workspace=ws)
            t = timeit(lambda: chain.getRegressorGram(q, dq, ddq, tm, layout="element", out=out, workspace=ws), reps=5, warm=2)
            if ref is None:
                ref = out[0].clone()
            print("%-18s %-18s %8.1f us -> %.3e evals/s   rel diff %.1e" % (name, label, t * 1e6, N / t, float((out[0] - ref).norm() / ref.norm())))
            for k in env:
                del os.environ[k]
