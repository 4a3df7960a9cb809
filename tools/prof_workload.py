#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 (kernel-trace / PMC passes): tools/prof_workload.py <gram|multi|stacked|torque>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain      # noqa: E402

which = sys.argv[1]
dev = torch.device("cuda:0")
F = os.path.join(ROOT, "tests", "fixtures")
G = (0, 0, -9.806)
if which == "gram":
    chain = Chain(os.path.join(F, "panda_like.urdf"), "link0", "link7", G)   # config 3: n=7, P=70
    n, N = 7, 4000000
    q, dq, ddq, tm = (torch.rand((n, N), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(4))
    for _ in range(4):
        chain.getRegressorGram(q, dq, ddq, tm, layout="element")                     # default: fused (pipelined LDS tile)
    for _ in range(4):
        chain.getRegressorGram(q, dq, ddq, tm, layout="element", chunk_samples=N)    # two kernels, image through HBM
elif which == "multi":
    from rosdyn_amd.multi import MultiChainRegressor
    from rosdyn_amd.urdf_gen import mixed_chain_set
    items = []
    for xml, base, tool in mixed_chain_set(F, 256):
        c = Chain(xml, base, tool, G)
        n = c.getActiveJointsNumber()
        items.append((c,) + tuple(torch.rand((n, 4096), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3)))
    plan = MultiChainRegressor(items)
    for _ in range(6):
        plan.run()
elif which == "stacked":
    chain = Chain(os.path.join(F, "ur10_like.urdf"), "base_link", "wrist_3_link", G)
    n, P, N = 6, 60, 1000000
    q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3))
    Y = torch.empty((P, N * n), dtype=torch.float64, device=dev)
    for _ in range(6):
        chain.getRegressor(q, dq, ddq, y_layout="stacked", out=Y)
elif which == "torque":
    chain = Chain(os.path.join(F, "ur10_like.urdf"), "base_link", "wrist_3_link", G)
    n, N = 6, 1000000
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3))
    for _ in range(6):
        chain.getJointTorque(q, dq, ddq, layout="element")
        chain.getJointInertia(q, layout="element")
        chain.getTransformations(q, layout="element")
        chain.getJacobian(q, layout="element")
        chain.getDTwist(q, dq, ddq, layout="element")
        chain.getWrench(q, dq, ddq, layout="element")
    T = chain.getTransformation(q, layout="element")
    seeds = q + 0.25 * (torch.rand_like(q) * 2 - 1)
    for _ in range(3):
        chain.computeLocalIk(T, seeds, toll=1e-6, max_iterations=30, layout="element")
torch.cuda.synchronize()
