"""Randomly doctored batches through the preconditioned route: whatever path the device takes, R must be numpy's Householder factor.
Each case: a random subset of joints moves eps x slower (or not at all) in a random set of tiles -- the subsample's tiles, a random
third of all tiles, or one contiguous stretch -- eps = 10^U(-12, 0); prints the path taken and the errors."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rosdyn_amd import Chain
from rosdyn_amd._lib import lib
from rosdyn_amd.samples import trajectory_batch
from oracle.oracle import OracleChain
from debug_cholqr3 import layout


def doctor(rng, q, dq, ddq, N, n):
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024); stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    tile_of = np.arange(N) // 16
    kind = rng.integers(0, 4)
    if kind == 0:
        sel = tile_of % stride == 0
    elif kind == 1:
        sel = rng.random(tiles)[tile_of] < 0.33
    elif kind == 2:
        a = rng.integers(0, tiles); b = a + rng.integers(1, tiles // 2)
        sel = (tile_of >= a) & (tile_of < b)
    else:
        sel = (tile_of % stride == 0) | (rng.random(tiles)[tile_of] < 0.1)
    joints = np.where(rng.random(n) < 0.6)[0]
    if len(joints) == 0:
        joints = np.array([rng.integers(0, n)])
    eps = 0.0 if rng.random() < 0.15 else 10.0 ** rng.uniform(-12, 0)
    idx = np.where(sel)[0]
    dq[idx[:, None], joints[None, :]] *= eps
    ddq[idx[:, None], joints[None, :]] *= eps
    if rng.random() < 0.3:                      # and the poses of those joints frozen as well
        q[idx[:, None], joints[None, :]] = q[idx[0], joints][None, :]
    return "kind %d joints %s eps %.1e" % (kind, joints.tolist(), eps)


def main(n_cases=24, N=66000, seed0=0, only=None):
    GRAV = (0, 0, -9.806)
    worst = 0.0
    for case in (range(n_cases) if only is None else only):
        rng = np.random.default_rng(seed0 + case)
        urdf, base, tool = [("ur10_like.urdf", "base_link", "wrist_3_link"), ("panda_like.urdf", "link0", "link7")][case % 2]
        path = os.path.join(ROOT, "tests/fixtures", urdf)
        chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
        n, P = ref.n, ref.P
        q, dq, ddq = trajectory_batch(seed0 + case, N, n)
        what = doctor(rng, q, dq, ddq, N, n)
        tau = ref.joint_torque(q, dq, ddq) + 1e-3 * rng.normal(size=(N, n))
        M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), tau.reshape(-1)])
        ws = torch.zeros((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h) // 8,), dtype=torch.float64, device="cuda")
        L, n1 = layout(n, ws.numel() * 8)
        R = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)), workspace=ws.view(torch.uint8)).cpu().numpy()
        ints = ws[L["flag"]:L["flag"] + 96].cpu().numpy().view(np.int32)
        G = M.T @ M
        s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
        s = np.linalg.svd(R, compute_uv=False)
        keep = s_ref > 1e-9 * s_ref[0]
        e1 = np.abs(R.T @ R - G).max() / np.abs(G).max()
        e2 = np.abs(s[keep] / s_ref[keep] - 1).max()
        e3 = (s[~keep].max() / s_ref[0]) if (~keep).any() else 0.0
        path_taken = "stand-by" if ints[1] else ("round 1" if ints[0] else ("round 0" if ints[2] else "stand-by (round 0 called off)"))
        worst = max(worst, e1, e2 * 1e-3)
        dg = ws[L["flag"] + 64:L["flag"] + 70].cpu().numpy()
        print(f"{case:3d} n={n} {what:55s} {path_taken:10s} R'R-G {e1:.1e}  sv {e2:.1e}  null {e3:.1e}   "
              f"gamma {dg[0]:.0e}/{dg[4]:.0e} rho {dg[2]:.3g} | gamma {dg[1]:.0e}/{dg[5]:.0e} rho {dg[3]:.3g}", flush=True)
    print("worst", worst)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 66000, int(sys.argv[3]) if len(sys.argv) > 3 else 0,
         [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else None)
