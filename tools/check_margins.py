import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle.oracle import OracleChain
from rosdyn_amd import Chain
from rosdyn_amd.samples import trajectory_batch
GRAV = (0.0, 0.0, -9.806)
path = "tests/fixtures/ur10_like.urdf"
chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
n, P, N = 6, 60, 330000
q, dq, ddq = trajectory_batch(99, N, n)
dq *= 1e-5; ddq *= 1e-5
tau = ref.joint_torque(q, dq, ddq)
M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), tau.reshape(-1)])
args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
R1 = chain.getRegressorTsqr(*args).cpu().numpy()
s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
s_gpu = np.linalg.svd(R1, compute_uv=False)
G, c, bb = chain.getRegressorGram(*args)
full = np.zeros((P + 1, P + 1)); full[:P, :P], full[:P, P], full[P, :P], full[P, P] = G.cpu().numpy(), c.cpu().numpy(), c.cpu().numpy(), float(bb.item())
s_ne = np.sqrt(np.abs(np.linalg.eigvalsh(full))[::-1])
keep = s_ref > 1e-11 * s_ref[0]
print("kept", keep.sum(), "min kept rel", s_ref[keep].min() / s_ref[0])
print("err_qr", np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max(), "err_ne", np.abs(s_ne[keep] / s_ref[keep] - 1.0).max())
# tau = None path on the new route
R0 = chain.getRegressorTsqr(args[0], args[1], args[2]).cpu().numpy()
print("no-tau: last col zero:", np.all(R0[:, P] == 0), "R'R vs G:", np.abs(R0[:P, :P].T @ R0[:P, :P] - full[:P, :P]).max() / np.abs(full).max())
