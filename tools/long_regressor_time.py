#!/usr/bin/env python3
"""Timing of getRegressor / getJointInertia / getJointTorque on chains with more than ten input joints (rdyn_long_local.hip): generated
all-revolute chains of 11, 14, 20 and 32 joints, all three regressor layouts."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rosdyn_amd import Chain                     # noqa: E402
from test_gpu_longkin import generated_revolute_chain  # noqa: E402
from tools.probe import timeit                   # noqa: E402

print("%-44s %10s %12s %10s" % ("call", "us / call", "B / eval", "GB/s"))
for nj, N in ((11, 200000), (14, 200000), (20, 100000), (32, 40000)):
    chain = Chain(generated_revolute_chain(nj, 1000 + nj), "l0", "l%d" % nj, (0, 0, -9.806))
    n, P = nj, 10 * nj
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
    qs, dqs, ddqs = (t.T.contiguous() for t in (q, dq, ddq))
    B = 32 * n + 8 * n * P
    for name, fn, nb in (("regressor + tau, element-major", lambda: chain.getRegressor(q, dq, ddq, layout="element", with_torque=True), B),
                         ("regressor + tau, per-sample images", lambda: chain.getRegressor(qs, dqs, ddqs, with_torque=True), B),
                         ("regressor + tau, stacked", lambda: chain.getRegressor(qs, dqs, ddqs, y_layout="stacked", with_torque=True), B),
                         ("joint inertia, element-major", lambda: chain.getJointInertia(q, layout="element"), 8 * n + 8 * n * n),
                         ("joint torque (wrench recursion)", lambda: chain.getJointTorque(q, dq, ddq, layout="element"), 32 * n)):
        t = timeit(fn, reps=5, warm=2)
        print("%2d joints, N = %6d: %-34s %10.1f %12d %10.0f" % (nj, N, name, t * 1e6, nb, nb * N / t / 1e9))
