#!/usr/bin/env python3
"""Placement experiment (round 6): the stacked / per-sample regressor launch (n = 6, N = 1e6) into ten output allocations of one process, with
the workgroups' chunk permutation off and on (RDYN_IMAGE_SCATTER = multiplier; tools only: the shipped default is decided from this table)."""
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
muls = [int(x) for x in (sys.argv[1:] or ["0", "7", "257", "4099"])]


def run(Y, layout):
    def f():
        c.getRegressor(q, dq, ddq, y_layout=layout, out=Y, tau_out=tau)
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


keep = []
for layout, shape in (("stacked", (P, N * n)), ("per_sample", (N, P, n))):
    print(layout, "us per launch; columns = RDYN_IMAGE_SCATTER", muls)
    for i in range(10):
        Y = torch.empty(shape, dtype=torch.float64, device="cuda")
        row = []
        for m in muls:
            if m:
                os.environ["RDYN_IMAGE_SCATTER"] = str(m)
            else:
                os.environ.pop("RDYN_IMAGE_SCATTER", None)
            row.append(run(Y, layout))
        os.environ.pop("RDYN_IMAGE_SCATTER", None)
        print("alloc %2d  " % i + "  ".join("%6.1f" % t for t in row))
        keep.append(Y)
        if i % 3 == 2:
            keep.append(torch.empty((1234567 * (i + 1),), dtype=torch.uint8, device="cuda"))
    keep = []
    torch.cuda.empty_cache()
