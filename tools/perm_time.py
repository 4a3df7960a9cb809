#!/usr/bin/env python3
"""Permuted input joints (6 joints, N = 1e6): per-sample images against the chain-order chain (k_image_sweep<.., PERM>)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                 # noqa: E402
from tools.probe import timeit               # noqa: E402

N, n = 1000000, 6
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
perm = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
names = perm.getMoveableJointNames()
order = [2, 0, 5, 1, 4, 3]
perm.setInputJointsName([names[i] for i in order])
qs, dqs, ddqs = (torch.rand((N, n), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
for name, c in (("chain order", chain), ("permuted", perm)):
    t = timeit(lambda: c.getRegressor(qs, dqs, ddqs, with_torque=True), reps=8, warm=2)
    print("%-12s per-sample images %8.1f us" % (name, t * 1e6), flush=True)
# same numbers: the permuted chain fed the permuted inputs returns the chain-order image with its rows permuted
idx = torch.tensor(order, device="cuda")
M = 4099
Y0, t0 = chain.getRegressor(qs[:M].contiguous(), dqs[:M].contiguous(), ddqs[:M].contiguous(), with_torque=True)
Yp, tp = perm.getRegressor(qs[:M][:, idx].contiguous(), dqs[:M][:, idx].contiguous(), ddqs[:M][:, idx].contiguous(), with_torque=True)
Y0 = Y0.reshape(M, -1, n) if Y0.dim() == 2 else Y0
print("shapes", tuple(Y0.shape), tuple(Yp.shape), "max |Y diff|", float((Yp.reshape(Y0.shape[0], -1, n) - Y0.reshape(Y0.shape[0], -1, n)[:, :, idx]).abs().max()),
      "max |tau diff|", float((tp - t0[:, idx]).abs().max()))
