#!/usr/bin/env python3
"""Placement experiment: time of the stacked / per-sample regressor launch against the virtual address of the output allocation
(14 allocations kept alive, holes of odd sizes between some of them)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain  # noqa: E402

N, n, P = 1000000, 6, 60
c = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
q, dq, ddq = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
tau = torch.empty((N, n), dtype=torch.float64, device="cuda")


def run(Y, layout):
    def f():
        c.getRegressor(q, dq, ddq, y_layout=layout, out=Y, tau_out=tau)
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8 * 1e3


keep = []
for i in range(14):
    Y = torch.empty((P * N * n,), dtype=torch.float64, device="cuda")
    ts = run(Y.view(P, N * n), "stacked")
    tp = run(Y.view(N, P, n), "per_sample")
    Y.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        Y.zero_()
    e1.record()
    torch.cuda.synchronize()
    tz = e0.elapsed_time(e1) / 8 * 1e3
    a = Y.data_ptr()
    print("alloc %2d  ptr 0x%012x  (ptr >> 21) & 0x3ff = %4d  (ptr>>30) = %5d   stacked %6.1f  per-sample %6.1f  fill %6.1f us" % (i, a, (a >> 21) & 0x3ff, a >> 30, ts, tp, tz))
    keep.append(Y)
    if i % 3 == 2:
        keep.append(torch.empty((1234567 * (i + 1),), dtype=torch.uint8, device="cuda"))
