#!/usr/bin/env python3
"""Panda-class arm (7 joints) with 7 friction components, N = 1e6: the identification Gram and R factor, time per call
(k_regressor_pgram_solo)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                 # noqa: E402
from rosdyn_amd.components import ComponentSet  # noqa: E402
from tools.probe import timeit               # noqa: E402

N, n7, E = 1000000, 7, "element"
q7, dq7, ddq7 = (torch.rand((n7, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
comps7 = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(n7)], n7)
for tool in ("link7", "hand"):
    pa = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", tool, (0, 0, -9.806))
    tau7 = pa.getJointTorque(q7, dq7, ddq7, layout=E)
    for name, fn in (("R factor of [A | tau]", lambda: pa.getRegressorTsqr(q7, dq7, ddq7, tau7, layout=E)),
                     ("identification Gram, 7 friction comps", lambda: pa.getIdentificationGram(comps7, q7, dq7, ddq7, tau7, layout=E)),
                     ("identification R factor, 7 friction comps", lambda: pa.getIdentificationTsqr(comps7, q7, dq7, ddq7, tau7, layout=E))):
        t = timeit(fn, reps=8, warm=2)
        print("panda link0->%-6s %-44s %9.1f us" % (tool, name, t * 1e6), flush=True)
