#!/usr/bin/env python3
"""GPU probe: per-sample regressor images at padded strides (alignment classes of the copy-out), 7-joint generated chain."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain
from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
from test_gpu_longchain import _chain_xml
N = 1000000
for nj, pads in ((7, (0, 6, 22)), (6, (0, 8, 24))):
    c = Chain(_chain_xml(nj, 100 + nj), "l0", "l%d" % nj, (0, 0, -9.806))
    n, P = nj, 10 * nj
    q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    for pad in pads:
        ss = n * P + pad
        buf = torch.empty((N * ss,), dtype=torch.float64, device="cuda")
        b = Batch(); b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, q.data_ptr(), dq.data_ptr(), ddq.data_ptr(), 0, 0
        b.stream = torch.cuda.current_stream().cuda_stream
        yl = RegressorLayout(ss, 1, n)
        f = lambda: check(lib().rdyn_regressor(c._h, C.byref(b), None, buf.data_ptr(), C.byref(yl)))
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        print("n=%d stride %d doubles = %d B (%.3f lines): %.1f us  %.0f GB/s of image bytes" % (n, ss, ss * 8, ss * 8 / 128.0, t * 1e6, N * n * P * 8 / t / 1e9))
        del buf
