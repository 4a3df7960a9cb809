#!/usr/bin/env python3
"""GPU probe: rdyn_regressor_tsqr vs rdyn_regressor_gram at the config 2 / config 3 sizes."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
for name, urdf, base, tool, N in (("cfg2 n=6 P=60", "ur10_like.urdf", "base_link", "wrist_3_link", 1000000), ("cfg3 n=7 P=70", "panda_like.urdf", "link0", "link7", 4000000)):
    c = Chain(os.path.join(ROOT, "tests/fixtures", urdf), base, tool, (0, 0, -9.806))
    n = c.getActiveJointsNumber()
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(4))
    for what in ("tsqr", "gram"):
        f = (lambda: c.getRegressorTsqr(q, dq, ddq, tau)) if what == "tsqr" else (lambda: c.getRegressorGram(q, dq, ddq, tau))
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        print("%s %s: %.3f ms" % (name, what, (time.perf_counter() - t0) / 3 * 1e3))
