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
midc = Chain(os.path.join(ROOT, "tests/fixtures/ur10_public_long.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))  # fixed head + fixed middle
skip = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", "link7", (0, 0, -9.806))
pn = skip.getActiveJointsName()
skip.setInputJointsName([pn[i] for i in (0, 1, 2, 4, 5, 6)])   # a moving joint in the middle left out
p7 = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", "link7", (0, 0, -9.806))
p7.setInputJointsName(list(reversed(pn)))
q7, dq7, ddq7 = (torch.rand((N, 7), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
longc = Chain(os.path.join(ROOT, "tests/fixtures/ur10_public_long.urdf"), "base_link", "tcp", (0, 0, -9.806))   # 14 joints, 6 inputs: expanded images
for name, c, args in (("14 joints, 6 inputs (P = 140): expanded images", longc, (qs, dqs, ddqs)), ("6 joints, chain order", chain, (qs, dqs, ddqs)), ("6 joints, permuted", perm, (qs, dqs, ddqs)),
                      ("8 joints, fixed head + fixed middle (P = 80)", midc, (qs, dqs, ddqs)),
                      ("7 joints, one left out mid-chain (6 x 70)", skip, (qs, dqs, ddqs)),
                      ("7 joints, reversed", p7, (q7, dq7, ddq7))):
    t = timeit(lambda: c.getRegressor(*args, with_torque=True), reps=8, warm=2)
    Yb = c.getActiveJointsNumber() * 10 * c.getLinksNumber() * 8 if hasattr(c, "getLinksNumber") else 0
    print("%-48s per-sample images %8.1f us" % (name, t * 1e6), flush=True)
# same numbers: the permuted chain fed the permuted inputs returns the chain-order image with its rows permuted
idx = torch.tensor(order, device="cuda")
M = 4099
Y0, t0 = chain.getRegressor(qs[:M].contiguous(), dqs[:M].contiguous(), ddqs[:M].contiguous(), with_torque=True)
Yp, tp = perm.getRegressor(qs[:M][:, idx].contiguous(), dqs[:M][:, idx].contiguous(), ddqs[:M][:, idx].contiguous(), with_torque=True)
Y0 = Y0.reshape(M, -1, n) if Y0.dim() == 2 else Y0
print("shapes", tuple(Y0.shape), tuple(Yp.shape), "max |Y diff|", float((Yp.reshape(Y0.shape[0], -1, n) - Y0.reshape(Y0.shape[0], -1, n)[:, :, idx]).abs().max()),
      "max |tau diff|", float((tp - t0[:, idx]).abs().max()))
