#!/usr/bin/env python3
"""GPU probe: fused regressor->Gram kernel vs the two-kernel path, configs 2 and 3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain          # noqa: E402
from rosdyn_amd._lib import lib       # noqa: E402
from tools.probe import timeit        # noqa: E402

for name, urdf, base, tool, N in (("cfg2", "ur10_like.urdf", "base_link", "wrist_3_link", 1000000),
                                  ("cfg3", "panda_like.urdf", "link0", "link7", 4000000)):
    chain = Chain(os.path.join(ROOT, "tests/fixtures", urdf), base, tool, (0, 0, -9.806))
    n = chain.getActiveJointsNumber()
    q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    ws = torch.empty((lib().rdyn_regressor_gram_workspace_bytes(chain._h, 0),), dtype=torch.uint8, device="cuda")
    out = chain.getRegressorGram(q, dq, ddq, tm, layout="element", workspace=ws)
    t = timeit(lambda: chain.getRegressorGram(q, dq, ddq, tm, layout="element", out=out, workspace=ws), reps=5, warm=2)
    print(name, "fused (RDYN_FUSED_BLOCKS=%s): %.1f us -> %.3e evals/s" % (os.environ.get("RDYN_FUSED_BLOCKS", "default"), t * 1e6, N / t))
