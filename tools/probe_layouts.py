#!/usr/bin/env python3
"""GPU probe: getRegressor in the three layouts for generated chains of 4..8 joints (per-sample image sizes with and without
partial lines: 8 joints = 5 120-byte images = 40 whole lines)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain
from test_gpu_longchain import _chain_xml
N = 1000000
for nj in (4, 6, 7, 8):
    c = Chain(_chain_xml(nj, 100 + nj), "l0", "l%d" % nj, (0, 0, -9.806))
    n, P = c.getActiveJointsNumber(), 10 * c.getJointsNumber()
    q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    qe, dqe, ddqe = (x.t().contiguous() for x in (q, dq, ddq))
    B = (4 * n + n * P) * 8
    for lay in ("per_sample", "stacked", "element"):
        shape = {"element": (P, n, N), "stacked": (P, N * n), "per_sample": (N, P, n)}[lay]
        Y = torch.empty(shape, dtype=torch.float64, device="cuda")
        args = (qe, dqe, ddqe) if lay == "element" else (q, dq, ddq)
        f = lambda: c.getRegressor(*args, layout="element" if lay == "element" else "sample", y_layout=lay, out=Y, with_torque=True)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        print("n=%d P=%d image %5d B (%.2f lines)  %-10s %7.1f us  %6.0f GB/s" % (n, P, n * P * 8, n * P * 8 / 128.0, lay, t * 1e6, B * N / t / 1e9))
        del Y
