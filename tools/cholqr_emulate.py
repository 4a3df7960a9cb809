"""numpy emulation of the preconditioned route's dense steps (k_cholqr_precond / k_cholqr_factor of rdyn_cholqr.hip) on the oracle's rows:
which columns are deferred / skipped and why, round by round.  CPU only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rosdyn_amd.samples import trajectory_batch
from oracle.oracle import OracleChain


def precond(R1, scale):
    n1 = R1.shape[0]
    A0 = np.triu(R1) * scale
    nrm = np.linalg.norm(A0, axis=0)
    mx = np.maximum.accumulate(nrm)
    piv = np.abs(np.diag(A0))
    floor = np.where(mx > 0, 1e-13 * mx, 1.0)
    residue = ~(piv >= floor)
    tiny = (piv < 1e-5 * nrm) & (np.arange(n1) + 1 < n1)
    z = residue | tiny
    lift = np.where(residue, floor, piv)
    kept = np.where(~z)[0]
    Qk, Rk = np.linalg.qr(A0[:, kept])
    T = np.zeros((n1, n1))
    T[np.ix_(kept, kept)] = Rk
    for k in np.where(z)[0]:
        coeff = Qk.T @ A0[:, k]
        left = kept < k
        T[kept[left], k] = coeff[left]
        T[k, k] = lift[k]
    W = np.linalg.inv(T)
    g = np.abs(W).T @ nrm
    gamma = np.max(np.where(nrm > 0, (np.abs(T).T @ g) / np.where(nrm > 0, nrm, 1), 0))
    return T, W, z, gamma, residue, tiny


def precond_gram(Gs, scale, rel_min=1e-10, floor_rel=1e-13):
    """preconditioner from the GRAM matrix of the subsample: Cholesky with deferred pivots (not used for elimination)"""
    n1 = Gs.shape[0]
    M = Gs * scale * scale
    g0 = np.diag(M).copy()
    nrm = np.sqrt(np.maximum(g0, 0))
    mx = np.maximum.accumulate(nrm)
    T = np.zeros((n1, n1))
    z = np.zeros(n1, bool)
    for k in range(n1):
        d = M[k, k]
        with np.errstate(all="ignore"):
            rel = d / g0[k]
        last = k + 1 == n1
        if not (rel >= (1e-14 if last else rel_min)) or not (d >= (floor_rel * mx[k]) ** 2):
            z[k] = True
            T[k, k] = floor_rel * mx[k] if mx[k] > 0 else 1.0
            continue
        piv = np.sqrt(d)
        T[k, k] = piv
        T[k, k + 1:] = M[k, k + 1:] / piv
        M[k + 1:, k + 1:] -= np.outer(T[k, k + 1:], T[k, k + 1:])
    W = np.linalg.inv(T)
    g = np.abs(W).T @ nrm
    gamma = np.max(np.where(nrm > 0, (np.abs(T).T @ g) / np.where(nrm > 0, nrm, 1), 0))
    return T, W, z, gamma, z.copy(), np.zeros(n1, bool)


def factor(G2, T, z, has_b=True):
    n1 = G2.shape[0]
    M = G2.copy()
    g0 = np.diag(M).copy()
    skip = np.zeros(n1, bool)
    rels = np.zeros(n1); pivs = np.zeros(n1)
    flag = 0
    R2 = np.zeros((n1, n1))
    for k in range(n1):
        d = M[k, k]
        with np.errstate(all="ignore"):
            rel = d / g0[k]
        piv = np.sqrt(max(d, 1e-30))
        rels[k], pivs[k] = rel, piv
        if z[k]:
            skip[k] = not (piv >= 0.1 and rel >= 1e-12)
            if piv >= 0.1 and not (rel >= 1e-12):
                flag = 1
        elif not (rel >= 1e-12):
            skip[k] = True
            flag = 1
        if skip[k]:
            continue
        R2[k, k] = piv
        R2[k, k + 1:] = M[k, k + 1:] / piv
        M[k + 1:, k + 1:] -= np.outer(R2[k, k + 1:], R2[k, k + 1:])
    kept = np.where(~skip)[0]
    Re = R2[np.ix_(kept, kept)] / np.sqrt(g0[kept])[None, :]
    rho = np.linalg.norm(np.linalg.inv(Re)) / np.sqrt(len(kept)) if len(kept) else 1.0
    R = R2 @ T
    nrm = np.linalg.norm(R, axis=0)
    W = np.linalg.inv(T)
    g = np.abs(W).T @ nrm
    gamma = np.max(np.where(nrm > 0, (np.abs(T).T @ g) / np.where(nrm > 0, nrm, 1), 0))
    if not rho <= 4.0 or not gamma <= 1e4:
        flag = 1
    return R, skip, rels, pivs, rho, flag, gamma


def run(M, sub_rows, tiles, sub_tiles, verbose=True, gram=False):
    n1 = M.shape[1]
    R1 = np.linalg.qr(M[sub_rows], mode="r")
    R1 = R1 * np.sign(np.diag(R1))[:, None] if False else R1
    scale = np.sqrt(tiles / sub_tiles)
    out = None
    for rnd in range(2):
        if gram and rnd == 0:
            Ms = M[sub_rows]
            T, W, z, gamma, residue, tiny = precond_gram(Ms.T @ Ms, scale)
        else:
            T, W, z, gamma, residue, tiny = precond(R1 if rnd == 0 else out, scale if rnd == 0 else 1.0)
        Q = M @ W
        G2 = Q.T @ Q
        R, skip, rels, pivs, rho, flag, gamma_all = factor(G2, T, z)
        if verbose:
            print(f"round {rnd}: gamma {gamma:.2e}  deferred {int(z.sum())} (residue {int(residue.sum())}, tiny {int((tiny & ~residue).sum())})  "
                  f"skipped {int(skip.sum())}  rho {rho:.3g}  gamma(all rows) {gamma_all:.2e}  flag {flag}")
            for k in range(n1):
                tag = ("R" if residue[k] else "t" if tiny[k] else " ") + ("S" if skip[k] else " ")
                if verbose > 1 and (z[k] or skip[k] or rels[k] < 0.25):
                    print(f"   col {k:2d} {tag} piv {pivs[k]:.2e} rel {rels[k]:.2e}")
        if gamma > 1e4:
            flag = 1
        out = R
        if not flag:
            return out
    print("-> stand-by (Householder of all rows)")
    return np.linalg.qr(M, mode="r")


if __name__ == "__main__":
    eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-9
    gram = len(sys.argv) > 2 and sys.argv[2] == "gram"
    slow = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    verbose = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    GRAV = (0, 0, -9.806)
    path = os.path.join(ROOT, "tests/fixtures/ur10_like.urdf")
    ref = OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, 330000
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024); stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    sub = (np.arange(N) // 16) % stride == 0
    q, dq, ddq = trajectory_batch(4711, N, n)
    dq *= slow; ddq *= slow
    dq[sub] *= eps; ddq[sub] *= eps
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(3).normal(size=(N, n))
    M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), tau.reshape(-1)])
    R = run(M, np.repeat(sub, n), tiles, (tiles + stride - 1) // stride, verbose=verbose, gram=gram)
    G = M.T @ M
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    s = np.linalg.svd(R, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    print("R'R-G", np.abs(R.T @ R - G).max() / np.abs(G).max(), "sv err", np.abs(s[keep] / s_ref[keep] - 1).max(), " smallest kept sv", s_ref[keep].min() / s_ref[0])
