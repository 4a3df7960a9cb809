#!/usr/bin/env python3
"""rocprofv3 workloads of round 3 (one per invocation, a few launches each):
  config5   BASELINE configs[4]: 256 mixed 6-/7-DOF chains x 4 096 samples through the plan, stacked layout (k_image_sweep_multi)
  cholqr    BASELINE configs[2] robust route: rdyn_regressor_tsqr at n = 7, N = 4e6 (k_regressor_tsqr subsample, k_regressor_pgram, small kernels)
  gram      rdyn_regressor_gram at config 2 and config 3 sizes (k_regressor_gram_duo)
  real      ur10_public base_link -> tool0 and panda link0 -> hand: stacked / per-sample regressor + Gram at N = 1e6"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain      # noqa: E402
G = (0, 0, -9.806)
what = sys.argv[1]
FX = os.path.join(ROOT, "tests", "fixtures")
if what == "config5":
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    items = []
    for xml, base, tool in mixed_chain_set(FX, 256):
        c = Chain(xml, base, tool, G)
        n = c.getActiveJointsNumber()
        items.append((c,) + tuple(torch.rand((n, 4096), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3)))
    plan = MultiChainRegressor(items, y_layout="stacked")
    for _ in range(4):
        plan.run()
    torch.cuda.synchronize()
elif what == "cholqr":
    c = Chain(os.path.join(FX, "panda_like.urdf"), "link0", "link7", G)
    N = 4000000
    q, dq, ddq, tm = (torch.rand((N, 7), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    for _ in range(3):
        c.getRegressorTsqr(q, dq, ddq, tm)
    torch.cuda.synchronize()
elif what == "gram":
    c = Chain(os.path.join(FX, "ur10_like.urdf"), "base_link", "wrist_3_link", G)
    q, dq, ddq, tm = (torch.rand((1000000, 6), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    for _ in range(4):
        c.getRegressorGram(q, dq, ddq, tm)
    c3 = Chain(os.path.join(FX, "panda_like.urdf"), "link0", "link7", G)
    q3, dq3, ddq3, tm3 = (torch.rand((4000000, 7), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    for _ in range(3):
        c3.getRegressorGram(q3, dq3, ddq3, tm3)
    torch.cuda.synchronize()
elif what == "real":
    for urdf, base, tool in (("ur10_public.urdf", "base_link", "tool0"), ("panda_like.urdf", "link0", "hand")):
        c = Chain(os.path.join(FX, urdf), base, tool, G)
        n, P, N = c.getActiveJointsNumber(), 10 * c.getJointsNumber(), 1000000
        q, dq, ddq, tm = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
        for lay, shape in (("stacked", (P, N * n)), ("per_sample", (N, P, n))):
            Y = torch.empty(shape, dtype=torch.float64, device="cuda")
            for _ in range(4):
                c.getRegressor(q, dq, ddq, y_layout=lay, out=Y)
            del Y
        for _ in range(4):
            c.getRegressorGram(q, dq, ddq, tm)
        torch.cuda.synchronize()
