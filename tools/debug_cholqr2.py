"""debug: unrepresentative subsample (static tiles) through the cholqr route; look for non-finite values stage by stage"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
def layout(nJ):
    n1 = 10 * nJ + 1; nb = (n1 + 15) // 16; nt = nb * (nb + 1) // 2
    off = ((256 + 128 + 2) * n1 * n1 + 31) & ~31
    L = {}
    for name, d in (("slabs", 256 * nt * 256), ("w", nt * 256), ("r_sub", n1 * n1), ("r1p", n1 * n1), ("g2", n1 * n1 + 1), ("r_swept", n1 * n1), ("v", n1 * n1), ("flag", 64)):
        L[name] = off; off = (off + d + 31) & ~31
    return L, n1
chain = Chain("tests/fixtures/ur10_like.urdf", "base_link", "wrist_3_link", (0, 0, -9.806))
n, N = 6, 330000
gen = torch.Generator(device="cuda").manual_seed(5)
q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(4))
tiles = (N + 15) // 16; stride = max(1, tiles // 2048)
sub = ((torch.arange(N, device="cuda") // 16) % stride == 0)
dq[sub] = 0; ddq[sub] = 0
L, n1 = layout(6)
ws = torch.zeros((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) // 8,), dtype=torch.float64, device="cuda")
R = chain.getRegressorTsqr(q, dq, ddq, tau, workspace=ws.view(torch.uint8))
torch.cuda.synchronize()
w = ws.cpu().numpy()
ints = w[L["flag"]:L["flag"] + 64].view(np.int32)
print("flag", ints[0], "zmask", ints[16:16 + n1].tolist())
for name in ("r_sub", "r1p", "w", "g2", "r_swept"):
    a = w[L[name]:L[name] + (n1 * n1 if name != "w" else 10 * 256)]
    print(name, "finite:", np.isfinite(a).all(), "max abs", np.nanmax(np.abs(a)))
Rn = R.cpu().numpy(); print("R finite", np.isfinite(Rn).all(), "nan count", np.isnan(Rn).sum())
T = w[L["r1p"]:L["r1p"] + n1 * n1].reshape(n1, n1).T
print("T diag", np.diag(T)[:30])
