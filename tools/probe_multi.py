#!/usr/bin/env python3
"""GPU probe: BASELINE.json configs[4] -- 256 distinct 6-/7-DOF chains x 4 096 samples each, one launch per joint-count group."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                         # noqa: E402
from rosdyn_amd.multi import MultiChainRegressor     # noqa: E402
from rosdyn_amd.urdf_gen import mixed_chain_set      # noqa: E402
from tools.probe import timeit                       # noqa: E402


def main(n_chains=256, S=4096):
    dev = torch.device("cuda:0")
    specs = mixed_chain_set(os.path.join(ROOT, "tests", "fixtures"), n_chains)
    items, nbytes = [], 0
    for xml, base, tool in specs:
        c = Chain(xml, base, tool, (0, 0, -9.806))
        n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
        q, dq, ddq = (torch.rand((n, S), dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3))
        items.append((c, q, dq, ddq))
        nbytes += S * (4 * n * 8 + n * P * 8)
    plan = MultiChainRegressor(items)
    t = timeit(plan.run, reps=10, warm=3)
    print("mixed batch: %d chains x %d samples = %d evals: %.1f us -> %.3e evals/s, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        n_chains, S, n_chains * S, t * 1e6, n_chains * S / t, nbytes / t / 1e9, nbytes / t / 8e12 * 100))


if __name__ == "__main__":
    main()
