#!/usr/bin/env python3
"""GPU probe: the stacked-regressor launch (n = 6, N = 1e6, 2.88 GB of output) at different byte offsets inside ONE big allocation:
is the HBM write rate a function of the output's base address?"""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
N = 1000000
c = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, P = 6, 60
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
tau = torch.empty((N, n), dtype=torch.float64, device="cuda")
big = torch.empty((P * N * n + (6 << 30) // 8,), dtype=torch.float64, device="cuda")
print("base 0x%x" % big.data_ptr())
def run(off_bytes):
    Y = big[off_bytes // 8: off_bytes // 8 + P * N * n].view(P, N * n)
    f = lambda: c.getRegressor(q, dq, ddq, y_layout="stacked", out=Y, tau_out=tau)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 6
for unit, count, name in ((4096, 17, "4 KB"), (1 << 16, 17, "64 KB"), (1 << 20, 33, "1 MB"), (1 << 25, 33, "32 MB"), (1 << 27, 40, "128 MB")):
    ts = [run(k * unit) * 1e6 for k in range(count)]
    print("step %-7s" % name, " ".join("%3.0f" % t for t in ts))
