#!/usr/bin/env python3
"""GPU probe: the stacked launch of several library builds (RDYN_LIBS=path:path:...) into the SAME 10 output allocations of one process
(each build gets its own dlopen'ed copy through ctypes)."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd._lib import Batch, RegressorLayout
N, n, P = 1000000, 6, 60
libs = []
for path in os.environ["RDYN_LIBS"].split(":"):
    L = C.CDLL(os.path.join(ROOT, path))
    h = C.c_void_p()
    g = (C.c_double * 3)(0, 0, -9.806)
    L.rdyn_chain_from_urdf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
    xml = open(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "rb").read()
    assert L.rdyn_chain_from_urdf(xml, b"base_link", b"wrist_3_link", g, C.byref(h)) == 0
    L.rdyn_regressor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    libs.append((os.path.basename(path), L, h))
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
yl = RegressorLayout(n, 1, N * n)
bufs = [torch.empty((P, N * n), dtype=torch.float64, device="cuda") for _ in range(10)]
for name, L, h in libs:
    ts = []
    for Y in bufs:
        b = Batch(); b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, q.data_ptr(), dq.data_ptr(), ddq.data_ptr(), 0, 0
        b.stream = torch.cuda.current_stream().cuda_stream
        f = lambda: L.rdyn_regressor(h, C.byref(b), None, Y.data_ptr(), C.byref(yl))
        assert f() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e6)
    print("%-22s" % name, " ".join("%4.0f" % t for t in ts))
