"""debug: run rdyn_regressor_tsqr on the cholqr route and look into the workspace (layout of rdyn_api.cpp: tsqr_layout)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
def layout(nJ):
    n1 = 10 * nJ + 1; nb = (n1 + 15) // 16; nt = nb * (nb + 1) // 2
    hh = (256 + 128 + 2) * n1 * n1
    off = (hh + 31) & ~31
    L = {}
    for name, d in (("slabs", 256 * nt * 256), ("w", nt * 256), ("r_sub", n1 * n1), ("r1p", n1 * n1), ("g2", n1 * n1 + 1), ("r_swept", n1 * n1), ("v", n1 * n1), ("flag", 64)):
        L[name] = off; off = (off + d + 31) & ~31
    L["total"] = off
    return L, n1
urdf, base, tool, N = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
chain = Chain(os.path.join("tests/fixtures", urdf), base, tool, (0, 0, -9.806))
n = chain.getActiveJointsNumber(); nJ = chain.getJointsNumber()
gen = torch.Generator(device="cuda").manual_seed(0x5EED0002)
q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(4))
nbytes = lib().rdyn_regressor_tsqr_workspace_bytes(chain._h)
L, n1 = layout(nJ)
print("ws bytes", nbytes, "layout total", L["total"] * 8)
ws = torch.zeros((nbytes // 8,), dtype=torch.float64, device="cuda")
R = chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws.view(torch.uint8))
torch.cuda.synchronize()
w = ws.cpu().numpy()
flag = w[L["flag"]:L["flag"] + 1].view(np.int32)[0]
Rsub = w[L["r_sub"]:L["r_sub"] + n1 * n1].reshape(n1, n1).T
R1p = w[L["r1p"]:L["r1p"] + n1 * n1].reshape(n1, n1).T
P = n1 - 1
G2 = w[L["g2"]:L["g2"] + P * P].reshape(P, P)
print("flag", flag)
print("diag Rsub", np.abs(np.diag(Rsub))[:12], "...")
print("G2 diag min/max", np.diag(G2).min(), np.diag(G2).max(), "offdiag max", np.abs(G2 - np.diag(np.diag(G2))).max())
G, c, bb = chain.getRegressorGram(q, dq, ddq, tau)
full = torch.zeros((P + 1, P + 1), dtype=torch.float64, device="cuda")
full[:P, :P], full[:P, P], full[P, :P], full[P, P] = G, c, c, bb[0]
print("R'R - G rel", ((R.t() @ R - full).abs().max() / full.abs().max()).item())
# G2 of round 1 against (M W)'(M W) from the GPU's own materialised regressor rows
if N <= 300000:
    Y, tau_g = chain.getRegressor(q, dq, ddq, with_torque=True)          # (N, P, n)
    M = torch.cat([Y.permute(0, 2, 1).reshape(N * n, P), tau.reshape(N * n, 1)], dim=1).cpu().numpy()
    W = np.linalg.inv(R1p)
    Q = M @ W
    G2ref = Q.T @ Q
    c2 = w[L["g2"] + P * P:L["g2"] + P * P + P]; bb2 = w[L["g2"] + P * P + P]
    full2 = np.zeros((P + 1, P + 1)); full2[:P, :P] = G2; full2[:P, P] = c2; full2[P, :P] = c2; full2[P, P] = bb2
    print("G2 vs host (M W)'(M W): max abs diff", np.abs(full2 - G2ref).max(), " |G2ref| max", np.abs(G2ref).max())
    d = np.abs(full2 - G2ref); i, j = np.unravel_index(np.argmax(d), d.shape); print("worst entry", i, j, full2[i, j], G2ref[i, j])
    print("W max", np.abs(W).max(), "R1p diag min", np.abs(np.diag(R1p)).min())
Rs = torch.from_numpy(Rsub.copy()).cuda()
print("sub factor scaled vs full: ", ((Rs.t() @ Rs) * (N / (2048 * 16)) - full).abs().max().item() / full.abs().max().item())
