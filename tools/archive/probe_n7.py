#!/usr/bin/env python3
"""GPU probe: why do 7-joint chains write the regressor at ~5.6 TB/s where 6- and 8-joint chains reach 6.3-6.8?  N sweep, element-major."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain
from test_gpu_longchain import _chain_xml
for nj in (6, 7, 8):
    c = Chain(_chain_xml(nj, 100 + nj), "l0", "l%d" % nj, (0, 0, -9.806))
    n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
    for N in (250000, 500000, 1000000, 1048576, 1500000):
        q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
        Y = torch.empty((P, n, N), dtype=torch.float64, device="cuda")
        f = lambda: c.getRegressor(q, dq, ddq, layout="element", y_layout="element", out=Y, with_torque=True)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        print("n=%d N=%8d  %7.1f us  %6.0f GB/s" % (n, N, t * 1e6, (4 * n + n * P) * 8 * N / t / 1e9))
        del Y, q, dq, ddq
