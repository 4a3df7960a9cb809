#!/usr/bin/env python3
"""rocprofv3 target: the identification R factor [Y | 7 friction columns | tau] of the 7-joint arm (panda_like link0 -> link7), N = 1e6, a few calls.
   cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 <repo>/tools/prof_ident.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain                    # noqa: E402
from rosdyn_amd.components import ComponentSet  # noqa: E402

N, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 7
tool = sys.argv[2] if len(sys.argv) > 2 else "link7"
chain = Chain(os.path.join(ROOT, "tests/fixtures/panda_like.urdf"), "link0", tool, (0, 0, -9.806))
q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda") * 2 - 1 for _ in range(3))
tau = chain.getJointTorque(q, dq, ddq, layout="element")
comps = ComponentSet([dict(type=0, joint=j, min_velocity=1e-3, max_velocity=10.0, parameters=[0.1, 0.2]) for j in range(n)], n)
for _ in range(6):
    chain.getIdentificationTsqr(comps, q, dq, ddq, tau, layout="element")
torch.cuda.synchronize()
if os.environ.get("RDYN_PROF_REPORT"):
    import ctypes as C
    from rosdyn_amd._lib import lib
    nbytes = lib().rdyn_identification_tsqr_workspace_bytes(chain._h, C.cast(comps._arr, C.c_void_p), comps.n_comps)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    chain.getIdentificationTsqr(comps, q, dq, ddq, tau, layout="element", workspace=ws)
    print("noise-free tau:", chain.lastTsqrReport(N, ws, components=comps))
    tau2 = tau + 1e-3 * torch.randn_like(tau)
    chain.getIdentificationTsqr(comps, q, dq, ddq, tau2, layout="element", workspace=ws)
    print("noisy tau:", chain.lastTsqrReport(N, ws, components=comps))
if os.environ.get("RDYN_PROF_TIME"):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        chain.getIdentificationTsqr(comps, q, dq, ddq, tau, layout="element")
    ev1.record()
    torch.cuda.synchronize()
    print("%.3f ms per call" % (ev0.elapsed_time(ev1) / 10))
