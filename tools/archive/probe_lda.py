#!/usr/bin/env python3
"""GPU probe: does the leading dimension of the stacked (N n) x P regressor (and the plane stride of the element-major image)
matter?  HBM channel / bank aliasing between the P (or n P) concurrently written column streams."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain
from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
from test_gpu_longchain import _chain_xml
N = int(os.environ.get("PROBE_N", "1000000"))
for nj in (6, 7):
    c = Chain(_chain_xml(nj, 100 + nj), "l0", "l%d" % nj, (0, 0, -9.806))
    n, P = nj, 10 * nj
    q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    tau = torch.empty((N, n), dtype=torch.float64, device="cuda")
    for kind in ("stacked", "element"):
        for pad in (0, 16, 64, 256, 1024, 4096, 16384, 65536 + 16, 262144 + 64):
            if kind == "stacked":
                lda = N * n + pad
                yl = RegressorLayout(n, 1, lda)
                size = P * lda
                qq, lay = (q, dq, ddq), 0
            else:
                plane = N + pad
                yl = RegressorLayout(1, plane, n * plane)
                size = P * n * plane
                qq, lay = tuple(x.t().contiguous() for x in (q, dq, ddq)), 1
            buf = torch.empty((size,), dtype=torch.float64, device="cuda")
            b = Batch(); b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, qq[0].data_ptr(), qq[1].data_ptr(), qq[2].data_ptr(), lay, 0
            b.stream = torch.cuda.current_stream().cuda_stream
            f = lambda: check(lib().rdyn_regressor(c._h, C.byref(b), None, buf.data_ptr(), C.byref(yl)))
            f(); f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): f()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            print("n=%d %-8s pad %7d doubles: %7.1f us  %6.0f GB/s" % (n, kind, pad, t * 1e6, (3 * n + n * P) * 8 * N / t / 1e9))
            del buf
