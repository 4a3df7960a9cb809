"""Which batches start the second round / the stand-by Householder call of the preconditioned route, and what comes out
(flags, deferred columns, gamma / rho of the two rounds read from the diagnostics at the end of the workspace: rdyn_api.cpp, tsqr_layout).
The subsample's tiles (every S-th) get velocities and accelerations scaled by eps: 0 = static (null columns in the subsample),
1e-9 .. 1e-4 = the subsample sees the inertia directions, but at a scale that says nothing about the rest of the batch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
from rosdyn_amd.samples import trajectory_batch
from oracle.oracle import OracleChain

def layout(nJ, workspace_bytes=None):
    """offsets (doubles) of the preconditioned route's regions in the factor workspace: a copy of tsqr_layout() in rdyn_api.cpp for chains
    without component columns (the supported way to read the outcome is rdyn_tsqr_last_report); checked against the library's size"""
    n1 = 10 * nJ + 1; nb = (n1 + 15) // 16; nt = nb * (nb + 1) // 2
    regions = (("slabs", 256 * nt * 256), ("w", nt * 256), ("r1p", n1 * n1), ("g2", n1 * n1 + 1), ("r_swept", n1 * n1), ("v", n1 * n1), ("r_full", 0), ("flag", 96), ("wide", 0))
    tail = sum((d + 31) & ~31 for _, d in regions)
    # the regions sit behind the Householder folds' own workspace (whichever of the register / LDS-resident folds needs more: its size is
    # the library's business): counted back from the end of the workspace the library asks for
    off = ((256 + 128 + 2) * n1 * n1 + 31) & ~31 if workspace_bytes is None else workspace_bytes // 8 - tail
    assert off >= 0 and off % 32 == 0, "tools/debug_cholqr3.py: layout() is out of step with rdyn_api.cpp"
    L = {}
    for name, d in regions:
        L[name] = off; off = (off + d + 31) & ~31
    assert workspace_bytes is None or workspace_bytes == off * 8, "tools/debug_cholqr3.py: layout() is out of step with rdyn_api.cpp"
    return L, n1

def main():
    GRAV = (0, 0, -9.806)
    path = os.path.join(ROOT, "tests/fixtures/ur10_like.urdf")
    chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, 330000
    ws = torch.zeros((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) // 8,), dtype=torch.float64, device="cuda")
    L, n1 = layout(6, ws.numel() * 8)
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024); stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    sub = (np.arange(N) // 16) % stride == 0
    for label, eps, joints in [("plain", None, None), ("static", 0.0, slice(None)), ("1e-9", 1e-9, slice(None)), ("1e-7", 1e-7, slice(None)), ("1e-5", 1e-5, slice(None)),
                               ("1e-3", 1e-3, slice(None)), ("j6 1e-9", 1e-9, slice(5, 6)), ("j6 1e-6", 1e-6, slice(5, 6)), ("j456 1e-8", 1e-8, slice(3, 6))]:
        q, dq, ddq = trajectory_batch(4711, N, n)
        if eps is not None:
            idx = np.where(sub)[0]
            dq[idx[:, None], np.arange(n)[joints][None, :]] *= eps
            ddq[idx[:, None], np.arange(n)[joints][None, :]] *= eps
        tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(3).normal(size=(N, n))
        M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), tau.reshape(-1)])
        args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
        ws.zero_()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        chain.getRegressorTsqr(*args, workspace=ws.view(torch.uint8))
        ev[0].record()
        R = chain.getRegressorTsqr(*args, workspace=ws.view(torch.uint8))
        ev[1].record()
        torch.cuda.synchronize()
        ints = ws[L["flag"]:L["flag"] + 96].cpu().numpy().view(np.int32)
        gam = ws[L["flag"] + 64:L["flag"] + 70].cpu().numpy()
        R = R.cpu().numpy()
        G = M.T @ M
        s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
        s_gpu = np.linalg.svd(R, compute_uv=False)
        keep = s_ref > 1e-9 * s_ref[0]
        print(f"{label:10s} flags {ints[0]} {ints[1]} {ints[2]}  null {int(ints[16:16 + n1].sum())}  {ev[0].elapsed_time(ev[1]):7.2f} ms  "
              f"R'R-G {np.abs(R.T @ R - G).max() / np.abs(G).max():.1e}  sv err {np.abs(s_gpu[keep] / s_ref[keep] - 1).max():.1e}  "
              f"null sv {s_gpu[~keep].max() / s_ref[0] if (~keep).any() else 0:.1e}  gamma {gam[0]:.1e} {gam[1]:.1e} rho {gam[2]:.2f} {gam[3]:.2f} gamma(all rows) {gam[4]:.1e} {gam[5]:.1e}", flush=True)


if __name__ == "__main__":
    main()
