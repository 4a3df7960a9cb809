#!/usr/bin/env python3
"""GPU probe: the stacked-regressor launch into buffers from hipExtMallocWithFlags (default / physically contiguous / uncached / fine-grained):
does an allocation flag pin the output-placement state (profiles/r2/placement.txt)?"""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
N = 1000000
c = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, P = 6, 60
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
size = P * N * n * 8
yl = RegressorLayout(n, 1, N * n)
def run(ptr):
    b = Batch(); b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, q.data_ptr(), dq.data_ptr(), ddq.data_ptr(), 0, 0
    b.stream = torch.cuda.current_stream().cuda_stream
    f = lambda: check(lib().rdyn_regressor(c._h, C.byref(b), None, ptr, C.byref(yl)))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 6 * 1e6
keep = []
for name, flag in (("default", 0), ("contiguous", 4), ("uncached", 3), ("finegrained", 1), ("default", 0), ("contiguous", 4)):
    ts = []
    for k in range(6):
        p = C.c_void_p()
        st = hip.hipExtMallocWithFlags(C.byref(p), size, flag)
        if st != 0:
            ts.append(-float(st)); continue
        ts.append(run(p.value))
        keep.append(p)
    print("%-12s" % name, " ".join("%4.0f" % t for t in ts))
