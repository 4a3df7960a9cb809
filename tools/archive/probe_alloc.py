#!/usr/bin/env python3
"""GPU probe: the same stacked-regressor launch (n = 6, N = 1e6) on output buffers at different addresses of one process."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
N = 1000000
c = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, P = 6, 60
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
tau = torch.empty((N, n), dtype=torch.float64, device="cuda")
keep = []
def run(Y):
    f = lambda: c.getRegressor(q, dq, ddq, y_layout="stacked", out=Y, tau_out=tau)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10
for i in range(12):
    Y = torch.empty((P, N * n), dtype=torch.float64, device="cuda")
    t = run(Y)
    a = Y.data_ptr()
    Y.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): Y.zero_()
    torch.cuda.synchronize()
    tz = (time.perf_counter() - t0) / 10
    print("alloc %2d  ptr 0x%012x  regressor %6.1f us  %5.0f GB/s   plain fill %6.1f us  %5.0f GB/s" % (i, a, t * 1e6, 3072 * N / t / 1e9, tz * 1e6, 2880 * N / tz / 1e9))
    keep.append(Y)
    if i % 3 == 2:  # hole of odd size in between
        keep.append(torch.empty((1234567 * (i + 1),), dtype=torch.uint8, device="cuda"))
# same buffers again, reverse order: is the time a property of the address?
for i in (9, 6, 3, 0):
    Y = [k for k in keep if k.dtype == torch.float64][i]
    print("again %2d ptr 0x%012x  %6.1f us" % (i, Y.data_ptr(), run(Y) * 1e6))
