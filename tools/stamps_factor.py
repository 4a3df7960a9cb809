"""phase stamps of k_cholqr_factor (a -DRDYN_CHOLQR_STAMPS build): 100 MHz wall clock ticks between the phases"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
from rosdyn_amd.samples import trajectory_batch
from debug_cholqr3 import layout
chain = Chain(os.path.join(ROOT, "tests/fixtures/ur10_like.urdf"), "base_link", "wrist_3_link", (0, 0, -9.806))
n, N = 6, 1000000
L, n1 = layout(6)
q, dq, ddq = trajectory_batch(1, N, n)
tau = np.random.default_rng(1).normal(size=(N, n))
args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
ws = torch.zeros((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) // 8,), dtype=torch.float64, device="cuda")
for _ in range(3):
    chain.getRegressorTsqr(*args, workspace=ws.view(torch.uint8))
torch.cuda.synchronize()
d = ws[L["flag"] + 62:L["flag"] + 69].cpu().numpy()
names = ["load", "cholesky", "rho, R2 into place, R = R2 T", "gamma"]
for i in range(4):
    print(f"{names[i + 1] if False else names[i]:10s} {(d[i + 1] - d[i]) / 100.0:8.1f} us")

p = ws[L["flag"] + 70:L["flag"] + 77].cpu().numpy()
for i, nm in enumerate(["load + norms", "cholesky", "T, V into place", "T out", "V out + gamma", "W in operand order"]):
    print(f"precond {nm:20s} {(p[i + 1] - p[i]) / 100.0:8.1f} us")
