#!/usr/bin/env python3
"""GPU probe: BASELINE configs[4] (256 mixed 6-/7-DOF chains x 4 096 samples) through the mixed-chain plan in the three Y layouts."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
from rosdyn_amd.multi import MultiChainRegressor
from rosdyn_amd.urdf_gen import mixed_chain_set
S = 4096
items, nbytes = [], 0
gen = torch.Generator(device="cuda").manual_seed(5)
for xml, base, tool in mixed_chain_set(os.path.join(ROOT, "tests", "fixtures"), 256):
    c = Chain(xml, base, tool, (0, 0, -9.806))
    n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
    items.append((c,) + tuple(torch.rand((n, S), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3)))
    nbytes += S * (4 * n + n * P) * 8
for rep in range(2):
    for lay in ("element", "stacked", "per_sample"):
        plan = MultiChainRegressor(items, y_layout=lay)
        plan.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-10s %7.1f us  %6.0f GB/s  %.3e evals/s" % (lay, ms * 1e3, nbytes / ms / 1e6, 256 * S / ms * 1e3))
        del plan
