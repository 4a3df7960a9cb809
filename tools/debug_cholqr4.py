"""pivots of the subsample factor relative to own / running-max column norms (which structural null is missed when the subsample is slow?)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
from rosdyn_amd.samples import trajectory_batch
from oracle.oracle import OracleChain
from debug_cholqr3 import layout
GRAV = (0, 0, -9.806)
path = os.path.join(ROOT, "tests/fixtures/ur10_like.urdf")
chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
n, P, N = 6, 60, 330000
L, n1 = layout(6)
ws = torch.zeros((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) // 8,), dtype=torch.float64, device="cuda")
tiles = (N + 15) // 16
stride = max(1, tiles // 1024); stride += 1 if (stride > 1 and stride % 2 == 0) else 0
sub = (np.arange(N) // 16) % stride == 0
for label, eps in [("plain", None), ("1e-3", 1e-3)]:
    q, dq, ddq = trajectory_batch(4711, N, n)
    if eps is not None:
        dq[sub] *= eps; ddq[sub] *= eps
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(3).normal(size=(N, n))
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    R = chain.getRegressorTsqr(*args, workspace=ws.view(torch.uint8))
    torch.cuda.synchronize()
    w = ws.cpu().numpy()
    Rs = w[L["r_sub"]:L["r_sub"] + n1 * n1].reshape(n1, n1).T
    zm = w[L["flag"]:L["flag"] + 64].view(np.int32)[16:16 + n1]
    nrm = np.linalg.norm(Rs, axis=0); M = np.maximum.accumulate(nrm)
    # numpy's view of the same subsample rows
    Msub = np.column_stack([ref.regressor(q[sub], dq[sub], ddq[sub]).reshape(-1, P), tau[sub].reshape(-1)])
    Rn = np.linalg.qr(Msub, mode="r")
    print(label, "validated null set:", np.where(zm)[0].tolist())
    print(" col  own-rel-pivot  max-rel-pivot  numpy-own-rel   null(precond rule)")
    for k in range(n1):
        d = abs(Rs[k, k])
        print(f" {k:3d}  {d / nrm[k]:.2e}  {d / M[k]:.2e}  {abs(Rn[k, k]) / np.linalg.norm(Rn[:, k]):.2e}  {'Z' if d < 1e-13 * M[k] else ''}")
