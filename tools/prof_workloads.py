#!/usr/bin/env python3
"""rocprofv3 workloads of the secondary legs (one per invocation, a few launches each; tools/gpu_profile.sh round):
  config5   BASELINE configs[4]: 256 mixed 6-/7-DOF chains x 4 096 samples through the plan, stacked layout (k_image_sweep_multi)
  cholqr    BASELINE configs[2] robust route: rdyn_regressor_tsqr at n = 7, N = 4e6 (k_regressor_tsqr subsample, k_regressor_pgram, small kernels)
  gram      rdyn_regressor_gram at config 2 and config 3 sizes (k_regressor_gram_duo)
  real      ur10_public base_link -> tool0 and panda link0 -> hand: stacked / per-sample regressor + Gram at N = 1e6
  ident     round 4: the identification R factor [Y | C | tau] -- panda link0 -> link7 + 7 friction components at N = 4e6 (k_regressor_pgram_solo),
            ur10_public base_link -> tool0 + 6 mixed components at N = 1e6 (reduced chain, k_regressor_pgram<6, XB>, k_cholqr_expand)
  long      round 4: the 14-joint fixture (ur10_public_long base_link -> tcp): element-major / per-sample regressor, Gram, R factor at N = 1e6
  tsqr_rows rdyn_tsqr on materialised matrices: 6e6 x 86 (k_pgram_rows<6>) and 1e6 x 112 (k_pgram_rows<7>, the dense steps in the workspace)"""
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
elif what == "ident":
    from rosdyn_amd.components import ComponentSet
    c = Chain(os.path.join(FX, "panda_like.urdf"), "link0", "link7", G)
    N = 4000000
    q, dq, ddq, tm = (torch.rand((N, 7), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(7)], 7)
    for _ in range(3):
        c.getIdentificationTsqr(comps, q, dq, ddq, tm)
    for _ in range(3):
        c.getIdentificationGram(comps, q, dq, ddq, tm)
    del q, dq, ddq, tm
    c = Chain(os.path.join(FX, "ur10_public.urdf"), "base_link", "tool0", G)
    N = 1000000
    q, dq, ddq, tm = (torch.rand((N, 6), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    comps = ComponentSet([dict(type=j % 3, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2, 0.01][:3 if j % 3 == 1 else 2]) for j in range(6)], 6)
    for _ in range(4):
        c.getIdentificationTsqr(comps, q, dq, ddq, tm)
    torch.cuda.synchronize()
elif what == "long":
    c = Chain(os.path.join(FX, "ur10_public_long.urdf"), "base_link", "tcp", G)
    N = 1000000
    q, dq, ddq, tm = (torch.rand((N, 6), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    qe, dqe, ddqe = (t.T.contiguous() for t in (q, dq, ddq))
    for _ in range(4):
        c.getRegressor(qe, dqe, ddqe, layout="element")
    for _ in range(3):
        c.getRegressor(q, dq, ddq)
    for _ in range(3):
        c.getRegressor(q, dq, ddq, y_layout="stacked")
    for _ in range(4):
        c.getRegressorGram(q, dq, ddq, tm)
        c.getRegressorTsqr(q, dq, ddq, tm)
    torch.cuda.synchronize()
elif what == "tsqr_rows":
    from rosdyn_amd.gram import tsqr
    for rows, cols in ((6000000, 85), (1000000, 111)):
        A = torch.rand((cols, rows), dtype=torch.float64, device="cuda")
        b = torch.rand((rows,), dtype=torch.float64, device="cuda")
        for _ in range(3):
            tsqr(A, b)
        del A, b
    torch.cuda.synchronize()
