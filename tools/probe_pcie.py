#!/usr/bin/env python3
"""PCIe-inclusive rate of the regressor for a caller that holds HOST buffers (pinned): H2D of (q, Dq, DDq), the kernel,
D2H of tau and the dense regressor.  The boundary itself takes device pointers; this is the number DESIGN.md section 6 quotes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain  # noqa: E402

N, n, P = 1000000, 6, 60
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
h_in = [torch.rand((N, n), dtype=torch.float64).pin_memory() * 2 - 1 for _ in range(3)]
h_in = [t.pin_memory() for t in h_in]
h_Y = torch.empty((P, N * n), dtype=torch.float64).pin_memory()
h_tau = torch.empty((N, n), dtype=torch.float64).pin_memory()
d_in = [torch.empty((N, n), dtype=torch.float64, device="cuda") for _ in range(3)]
d_Y = torch.empty((P, N * n), dtype=torch.float64, device="cuda")
d_tau = torch.empty((N, n), dtype=torch.float64, device="cuda")


def once():
    for d, h in zip(d_in, h_in):
        d.copy_(h, non_blocking=True)
    chain.getRegressor(d_in[0], d_in[1], d_in[2], y_layout="stacked", out=d_Y, tau_out=d_tau)
    h_tau.copy_(d_tau, non_blocking=True)
    h_Y.copy_(d_Y, non_blocking=True)
    torch.cuda.synchronize()


once()
t0 = time.perf_counter()
for _ in range(5):
    once()
dt = (time.perf_counter() - t0) / 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
h_Y.copy_(d_Y, non_blocking=True)
e1.record()
torch.cuda.synchronize()
d2h = e0.elapsed_time(e1) * 1e-3
print("host-resident caller: %.1f ms per 1e6 samples -> %.3e evals/s (D2H of the 2.88 GB regressor alone: %.1f ms = %.1f GB/s)" % (
    dt * 1e3, N / dt, d2h * 1e3, 2.88 / d2h))
