"""GPU tests of the tall-skinny QR path (rdyn_tsqr.hip; BASELINE.json configs[2] "regressor + TSQR"): the R factor of [A | b]
without forming A'A.  Oracle: numpy.linalg.qr of the same rows (|R| up to row signs where A has full column rank), the invariant
R1'R1 = [A b]'[A b] everywhere, and the case the Gram route cannot do: cond(A) = 1e9."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)


def _abs_rows_equal(R, Rq, tol):
    s = np.sign(np.diag(R)) * np.sign(np.diag(Rq))
    s[s == 0] = 1.0
    assert np.abs(R - s[:, None] * Rq).max() <= tol * np.abs(Rq).max()


@pytest.mark.parametrize("rows,n,with_b", [(1000, 12, True), (33, 5, False), (7, 15, True), (50001, 31, True), (4096, 63, True), (64, 16, False)])
def test_generic_tsqr_matches_numpy_qr(rows, n, with_b):
    torch = pytest.importorskip("torch")
    from rosdyn_amd.gram import tsqr
    rng = np.random.default_rng(rows + n)
    A = rng.normal(size=(rows, n))
    b = rng.normal(size=rows) if with_b else None
    At = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()
    R1 = tsqr(At, torch.from_numpy(b).cuda() if with_b else None).cpu().numpy()
    M = np.column_stack([A, b]) if with_b else A
    assert np.allclose(np.tril(R1, -1), 0.0)
    assert np.abs(R1.T @ R1 - M.T @ M).max() <= 1e-12 * np.abs(M.T @ M).max()
    if rows >= M.shape[1]:
        _abs_rows_equal(R1, np.linalg.qr(M, mode="r"), 1e-11)


def test_tsqr_solves_what_the_normal_equations_cannot():
    """cond(A) = 1e9 (cond(A'A) = 1e18 > 1 / eps): TSQR + rdyn_solve_r_factor recover x to ~cond * eps, the Gram route does not."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.gram import gram, solve_base_parameters, solve_r_factor, tsqr
    rng = np.random.default_rng(3)
    m, n = 20000, 24
    U, _ = np.linalg.qr(rng.normal(size=(m, n)))
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    A = U @ np.diag(np.logspace(0, -9, n)) @ V.T
    x_true = V @ rng.normal(size=n)
    b = A @ x_true
    At, bt = torch.from_numpy(np.ascontiguousarray(A.T)).cuda(), torch.from_numpy(b).cuda()
    R1 = tsqr(At, bt)
    x_qr, rank = solve_r_factor(R1, n, rtol=1e-13)
    assert rank == n
    err_qr = np.abs(x_qr - x_true).max() / np.abs(x_true).max()
    G, c, _ = gram(At, bt)
    x_ne, _ = solve_base_parameters(G, c, rtol=1e-15)
    err_ne = np.abs(x_ne - x_true).max() / np.abs(x_true).max()
    assert err_qr <= 1e-5, err_qr                # ~ cond * eps
    assert err_ne >= 1e-3 > 100 * err_qr, (err_ne, err_qr)


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "wrist_3_link"), ("ur10_like.urdf", "base_link", "tool0"),
                                            ("panda_like.urdf", "link0", "link7")], ids=["6of6", "6of7", "7of7_two_slots"])
@pytest.mark.parametrize("N", [1, 17, 1000])
def test_regressor_tsqr_against_the_oracle_rows(urdf, base, tool, N):
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import solve_r_factor
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    q, dq, ddq = trajectory_batch(40 + N, N, n)
    rng = np.random.default_rng(N)
    Y = ref.regressor(q, dq, ddq)                                  # (N, n, P)
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * rng.normal(size=(N, n))
    M = np.column_stack([Y.reshape(-1, P), tau.reshape(-1)])       # rows (s, j)
    R1 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))).cpu().numpy()
    assert R1.shape == (P + 1, P + 1) and np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()  # Q is orthogonal: the factor reproduces the normal equations
    if N == 1000:
        # least-squares solution straight from the factor == numpy lstsq on the oracle's rows (minimum norm: rank deficient)
        x, rank = solve_r_factor(R1, P, rtol=1e-9)
        x_ls = np.linalg.lstsq(M[:, :P], M[:, P], rcond=1e-9)[0]
        assert rank == np.linalg.matrix_rank(M[:, :P], tol=1e-9 * np.linalg.norm(M[:, :P], 2))
        assert np.abs(M[:, :P] @ (x - x_ls)).max() <= 1e-8 * np.abs(M[:, P]).max()
        # the residual norm from the factor alone: rho^2 + |R x - d|^2 (the second term: the rank-deficient directions)
        res = np.linalg.norm(M[:, :P] @ x_ls - M[:, P])
        res_f = np.sqrt(R1[P, P] ** 2 + np.sum((R1[:P, :P] @ x - R1[:P, P]) ** 2))
        assert abs(res_f - res) <= 1e-6 * res


def test_accumulate_and_host_combine_equal_one_shot():
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import tsqr_combine_host
    from rosdyn_amd.samples import trajectory_batch
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", GRAV)
    n, N = 6, 3000
    q, dq, ddq = (torch.from_numpy(x).cuda() for x in trajectory_batch(9, N, n))
    tau = chain.getJointTorque(q, dq, ddq)
    full = chain.getRegressorTsqr(q, dq, ddq, tau).cpu().numpy()
    h = 1234
    first = chain.getRegressorTsqr(q[:h].contiguous(), dq[:h].contiguous(), ddq[:h].contiguous(), tau[:h].contiguous())
    second = chain.getRegressorTsqr(q[h:].contiguous(), dq[h:].contiguous(), ddq[h:].contiguous(), tau[h:].contiguous())
    out = first.clone()
    acc = chain.getRegressorTsqr(q[h:].contiguous(), dq[h:].contiguous(), ddq[h:].contiguous(), tau[h:].contiguous(), out=out, accumulate=True)
    assert acc is out                                    # out= is written, not only returned
    ref = full.T @ full
    for R in (acc.cpu().numpy(), tsqr_combine_host([first.cpu().numpy(), second.cpu().numpy()])):
        assert np.allclose(np.tril(R, -1), 0.0)
        assert np.abs(R.T @ R - ref).max() <= 1e-11 * np.abs(ref).max()
    # reproducible: same launch, same bits
    again = chain.getRegressorTsqr(q, dq, ddq, tau).cpu().numpy()
    assert np.array_equal(full, again)


@pytest.mark.parametrize("urdf,base,tool", [("ur10_like.urdf", "base_link", "wrist_3_link"), ("mixed_joints.urdf", "world", "slider"), ("panda_like.urdf", "link0", "link5"),
                                            ("panda_like.urdf", "link0", "link7")], ids=["ur10_6", "mixed", "panda5", "panda7"])
@pytest.mark.parametrize("N", [1, 33, 2000])
def test_identification_tsqr_against_the_oracle_rows(urdf, base, tool, N):
    """[Y | C | tau_meas] with friction / spring columns folded straight into the R factor (rdyn_identification_tsqr): the factor
    reproduces the normal equations of the oracle's rows, accumulates over chunks, and its least-squares solution returns the
    friction coefficients the torques were made with."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.components import FRICTION1, FRICTION2, SPRING, ComponentSet
    from rosdyn_amd.gram import solve_r_factor
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    q, dq, ddq = trajectory_batch(70 + N, N, n)
    # one component per joint, the three kinds mixed (K = 2, 3, 2, 2, 3, ... columns)
    kinds = [FRICTION1, FRICTION2, SPRING]
    specs, dicts = [], []
    for j in range(n):
        ty = kinds[j % 3]
        par = [0.5 + 0.1 * j, 1.0 + 0.2 * j] + ([0.05] if ty == FRICTION2 else [])
        specs.append((ty, j, 1e-3, 5.0, par))
        dicts.append(dict(type=ty, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=par))
    comps = ComponentSet(dicts, n)
    K = comps.columns
    Cm, tau_c = components_regressor(specs, n, q, dq)               # (N, n, K), (N, n)
    rng = np.random.default_rng(N)
    Y = ref.regressor(q, dq, ddq)
    tau = ref.joint_torque(q, dq, ddq) + tau_c + 1e-3 * rng.normal(size=(N, n))
    M = np.column_stack([Y.reshape(-1, P), Cm.reshape(-1, K), tau.reshape(-1)])
    tq, tdq, tddq, ttau = (torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))
    R1 = chain.getIdentificationTsqr(comps, tq, tdq, tddq, ttau).cpu().numpy()
    n1 = P + K + 1
    assert R1.shape == (n1, n1) and np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()
    # element-major inputs give the same factor (same arithmetic: bit for bit)
    R1e = chain.getIdentificationTsqr(comps, *(x.t().contiguous() for x in (tq, tdq, tddq, ttau)), layout="element").cpu().numpy()
    assert np.array_equal(R1, R1e)
    if N == 2000:
        h = 777
        first = chain.getIdentificationTsqr(comps, *(x[:h].contiguous() for x in (tq, tdq, tddq, ttau)))
        acc = chain.getIdentificationTsqr(comps, *(x[h:].contiguous() for x in (tq, tdq, tddq, ttau)), out=first.clone(), accumulate=True).cpu().numpy()
        assert np.abs(acc.T @ acc - G).max() <= 1e-11 * np.abs(G).max()
        x, rank = solve_r_factor(R1, P + K, rtol=1e-9)
        x_ls = np.linalg.lstsq(M[:, :P + K], M[:, P + K], rcond=1e-9)[0]
        assert np.abs(M[:, :P + K] @ (x - x_ls)).max() <= 1e-8 * np.abs(M[:, P + K]).max()
        truth = np.concatenate([sp[4] for sp in specs])
        assert np.abs(x[P:] - truth).max() < 0.02                   # the component parameters are identifiable: they come back
    # without components it is the regressor TSQR
    R0 = chain.getIdentificationTsqr(None, tq, tdq, tddq, ttau).cpu().numpy()
    R0b = chain.getRegressorTsqr(tq, tdq, tddq, ttau).cpu().numpy()
    assert np.array_equal(R0, R0b)


@pytest.mark.parametrize("urdf,base,tool,N", [("ur10_like.urdf", "base_link", "wrist_3_link", 1000000), ("panda_like.urdf", "link0", "link7", 4000000)],
                         ids=["config2_size", "config3_size"])
def test_full_size_factor_reproduces_the_mfma_normal_equations(urdf, base, tool, N):
    """BASELINE configs[1] / configs[2] sizes: two independent reductions of the same 6e6 / 28e6 regressor rows -- Householder folds
    on the vector units (TSQR) and the fp64-MFMA Gram -- must agree: R1'R1 = [G c; c' bb]."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    chain = Chain(os.path.join(FIXTURES, urdf), base, tool, GRAV)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(4))
    R1 = chain.getRegressorTsqr(q, dq, ddq, tau)
    G, c, bb = chain.getRegressorGram(q, dq, ddq, tau)
    full = torch.zeros((P + 1, P + 1), dtype=torch.float64, device="cuda")
    full[:P, :P], full[:P, P], full[P, :P], full[P, P] = G, c, c, bb[0]
    err = (R1.t() @ R1 - full).abs().max().item()
    assert err <= 1e-10 * full.abs().max().item(), err
    assert torch.equal(torch.tril(R1, -1), torch.zeros_like(R1))


# ---- the preconditioned CholeskyQR route (rdyn_cholqr.hip): batches of >= 4 096 samples, the heavy pass on the matrix cores
CHOLQR_N = 330000
CHOLQR_CASES = [("ur10_like.urdf", "base_link", "wrist_3_link"),      # 6 joints: W in LDS beside the four tiles
                ("panda_like.urdf", "link0", "link7"),                # 7 joints: W read from global memory (four tiles fill the LDS)
                ("ur10_public.urdf", "base_link", "tool0"),           # fixed head + two fixed tail joints: reduced chain swept, factor expanded (P = 90)
                ("panda_like.urdf", "link0", "hand")]



def _workspace(chain):
    import torch
    from rosdyn_amd._lib import lib
    return torch.empty((lib().rdyn_regressor_tsqr_workspace_bytes(chain._h),), dtype=torch.uint8, device="cuda")


def _oracle_rows(ref, q, dq, ddq, tau):
    Y = ref.regressor(q, dq, ddq)
    return np.column_stack([Y.reshape(-1, ref.P), tau.reshape(-1)])


@pytest.mark.parametrize("urdf,base,tool", CHOLQR_CASES, ids=["ur10_6", "panda_7", "ur10_public_tool0", "panda_hand"])
def test_cholqr_route_against_numpy_qr_of_the_oracle_rows(urdf, base, tool):
    """N = 330 000 (the route of every batch of 4 096 samples or more): upper triangular, R'R = M'M, the singular values of R equal those of numpy's
    Householder factor of the oracle's rows, same minimum-norm least-squares solution; accumulation folds a second factor in."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import solve_r_factor
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P, N = ref.n, ref.P, CHOLQR_N
    q, dq, ddq = trajectory_batch(2024, N, n)
    rng = np.random.default_rng(7)
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * rng.normal(size=(N, n))
    M = _oracle_rows(ref, q, dq, ddq, tau)
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    ws = _workspace(chain)
    R1 = chain.getRegressorTsqr(*args, workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws)          # a trajectory batch: the first round is accepted (what the device measured: gamma, rho)
    assert rep["route"] == 1 and rep["stage"] == 0 and 0 < rep["gamma"][0] <= 1e4 and 0.99 <= rep["rho"][0] <= 1.5, rep
    assert R1.shape == (P + 1, P + 1) and np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()
    Rq = np.linalg.qr(M, mode="r")
    s_gpu, s_ref = np.linalg.svd(R1, compute_uv=False), np.linalg.svd(Rq, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9
    assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])
    x, rank = solve_r_factor(R1, P, rtol=1e-9)
    x_ls = np.linalg.lstsq(M[:, :P], M[:, P], rcond=1e-9)[0]
    assert rank == int(keep[: P + 1].sum()) - 1 or rank == np.linalg.matrix_rank(M[:, :P], tol=1e-9 * s_ref[0])
    assert np.abs(M[:, :P] @ (x - x_ls)).max() <= 1e-8 * np.abs(M[:, P]).max()
    # accumulate: the factor of twice the rows
    R2 = chain.getRegressorTsqr(*args, out=torch.from_numpy(R1).cuda(), accumulate=True).cpu().numpy()
    assert np.allclose(np.tril(R2, -1), 0.0)
    assert np.abs(R2.T @ R2 - 2 * G).max() <= 1e-11 * np.abs(G).max()
    # reproducible
    assert np.array_equal(R1, chain.getRegressorTsqr(*args).cpu().numpy())


def test_cholqr_route_keeps_small_singular_values():
    """A slow trajectory (velocities and accelerations scaled by 1e-5): ten gravity directions at O(1), the other 26 identifiable
    directions at 1e-6 .. 6e-8 of that -- cond(A) ~ 2e7 on the identifiable subspace, cond^2 eps ~ 0.03.  The normal equations lose the small singular values (error ~ cond^2 eps relative to themselves); the preconditioned
    route must return them like numpy's Householder QR of the same rows does."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, CHOLQR_N
    q, dq, ddq = trajectory_batch(99, N, n)
    dq *= 1e-5
    ddq *= 1e-5
    tau = ref.joint_torque(q, dq, ddq)
    M = _oracle_rows(ref, q, dq, ddq, tau)
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    R1 = chain.getRegressorTsqr(*args).cpu().numpy()
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1, compute_uv=False)
    G, c, bb = chain.getRegressorGram(*args)
    full = np.zeros((P + 1, P + 1))
    full[:P, :P], full[:P, P], full[P, :P], full[P, P] = G.cpu().numpy(), c.cpu().numpy(), c.cpu().numpy(), float(bb.item())
    s_ne = np.sqrt(np.abs(np.linalg.eigvalsh(full))[::-1])
    keep = s_ref > 1e-11 * s_ref[0]                      # everything a cond-1e11 factor can hold
    assert s_ref[keep].min() < 1e-6 * s_ref[0]           # the case is as ill conditioned as it claims
    err_qr = np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max()
    err_ne = np.abs(s_ne[keep] / s_ref[keep] - 1.0).max()
    assert err_qr <= 1e-5, (err_qr, err_ne)
    assert err_ne > 100 * err_qr, (err_qr, err_ne)


def test_cholqr_route_survives_an_unrepresentative_subsample():
    """The preconditioner comes from every S-th 16-sample tile.  Here exactly those tiles are STATIC poses (zero velocities and
    accelerations): the subsample only sees the gravity columns and its factor declares every inertia column null although the other
    90 % of the rows excite them.  The factor kernel must
    notice (their pivots are far above the rounding level), keep them and run the second round: same singular values as numpy's
    Householder factor of all rows."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, CHOLQR_N
    q, dq, ddq = trajectory_batch(4711, N, n)
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024)                       # rdyn_api.cpp: the subsample pass sweeps tiles 0, stride, 2 stride, ...
    stride += 1 if (stride > 1 and stride % 2 == 0) else 0   # (an even stride is made odd)
    sub = (np.arange(N) // 16) % stride == 0
    dq[sub] = 0.0
    ddq[sub] = 0.0
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(3).normal(size=(N, n))   # a residual: the last column counts too
    M = _oracle_rows(ref, q, dq, ddq, tau)
    ws = _workspace(chain)
    R1 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)), workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws)
    assert rep["route"] == 1 and rep["stage"] == 1 and rep["gamma"][0] > 1e4 and rep["gamma"][1] <= 1e4 and rep["rho"][1] <= 4.0, rep
    assert np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9
    assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])
    # and the subsample really was blind: its own rows have a smaller rank than the whole batch
    assert np.linalg.matrix_rank(M[np.repeat(sub, n), :P], tol=1e-9 * s_ref[0]) < np.linalg.matrix_rank(M[:, :P], tol=1e-9 * s_ref[0])


@pytest.mark.parametrize("eps,joints,stage", [(1e-3, slice(None), 1), (1e-7, slice(None), 1), (1e-9, slice(None), 2), (1e-8, slice(3, 6), 0)],
                         ids=["1e-3", "1e-7", "1e-9_standby", "joints456_1e-8"])
def test_cholqr_route_with_a_slow_subsample(eps, joints, stage):
    """The tiles the preconditioner is built from move 1e-3 .. 1e-9 times slower than the rest of the batch: the subsample sees every
    direction, but at scales that say nothing about the batch -- its inverse factor has entries that are harmless on the subsample's own
    rows and amplify the rounding of Q = A W by up to 1e9 on the others (round 2 was fine with itself and R'R - G was 1e-6 before the
    acceptance test measured that growth on the norms of all rows).  Whatever path the device takes -- second round, or, for the 1e-9
    case, the stand-by Householder factorisation of all rows -- the factor must be that of numpy's Householder QR."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, CHOLQR_N
    q, dq, ddq = trajectory_batch(4711, N, n)
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024)
    stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    idx = np.where((np.arange(N) // 16) % stride == 0)[0]
    cols = np.arange(n)[joints]
    dq[idx[:, None], cols[None, :]] *= eps
    ddq[idx[:, None], cols[None, :]] *= eps
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * np.random.default_rng(3).normal(size=(N, n))
    M = _oracle_rows(ref, q, dq, ddq, tau)
    ws = _workspace(chain)
    R1 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)), workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws)
    assert rep["route"] == 1 and rep["stage"] == stage, rep      # second round / stand-by / first round: decided on the device
    assert np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-13 * np.abs(G).max()
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-10
    assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])


@pytest.mark.parametrize("urdf,base,tool", [CHOLQR_CASES[0], CHOLQR_CASES[1], CHOLQR_CASES[2]], ids=["ur10_6", "panda_7", "ur10_public_tool0"])
@pytest.mark.parametrize("N", [4096, 4097, 5555, 20000, 40000])
def test_cholqr_route_from_its_threshold_on(urdf, base, tool, N):
    """The route's smallest batches (4 096 samples = 256 tiles: the "subsample" is the whole batch up to 2 047 tiles, every second or
    third tile after that), element-major inputs, with and without a measured torque, the layouts of the batch, accumulation into a
    factor that came from the other route (N = 1 000: Householder folds)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P = ref.n, ref.P
    q, dq, ddq = trajectory_batch(N, N, n)
    tau = ref.joint_torque(q, dq, ddq) + 1e-2 * np.random.default_rng(N).normal(size=(N, n))
    M = _oracle_rows(ref, q, dq, ddq, tau)
    G = M.T @ M
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    dev = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    elem = [x.t().contiguous() for x in dev]
    for args, layout in ((dev, "sample"), (elem, "element")):
        R1 = chain.getRegressorTsqr(*args, layout=layout).cpu().numpy()
        assert np.allclose(np.tril(R1, -1), 0.0)
        assert np.abs(R1.T @ R1 - G).max() <= 1e-12 * np.abs(G).max()
        s_gpu = np.linalg.svd(R1, compute_uv=False)
        assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9
        assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])
    # no measured torque: the last column (and row) of the factor is zero
    R0 = chain.getRegressorTsqr(*dev[:3]).cpu().numpy()
    assert np.all(R0[:, P] == 0.0) and np.abs(R0[:P, :P].T @ R0[:P, :P] - G[:P, :P]).max() <= 1e-12 * np.abs(G).max()
    # a factor of 1 000 other samples (Householder route) accumulated into
    q2, dq2, ddq2 = trajectory_batch(N + 1, 1000, n)
    tau2 = ref.joint_torque(q2, dq2, ddq2)
    M2 = _oracle_rows(ref, q2, dq2, ddq2, tau2)
    ws = _workspace(chain)
    Rs = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q2, dq2, ddq2, tau2)), workspace=ws)
    assert chain.lastTsqrReport(1000, ws)["route"] == 0
    Ra = chain.getRegressorTsqr(*dev, out=Rs.contiguous(), accumulate=True).cpu().numpy()
    G2 = G + M2.T @ M2
    assert np.allclose(np.tril(Ra, -1), 0.0) and np.abs(Ra.T @ Ra - G2).max() <= 1e-12 * np.abs(G2).max()


def _doctor(rng, q, dq, ddq, N, n):
    """a random subset of joints moves eps x slower (or not at all, or is frozen) in a random set of 16-sample tiles: the tiles the
    preconditioner is built from, a random third of all tiles, one contiguous stretch, or a mixture"""
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024)
    stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    tile_of = np.arange(N) // 16
    kind = rng.integers(0, 4)
    if kind == 0:
        sel = tile_of % stride == 0
    elif kind == 1:
        sel = rng.random(tiles)[tile_of] < 0.33
    elif kind == 2:
        a = rng.integers(0, tiles)
        sel = (tile_of >= a) & (tile_of < a + rng.integers(1, tiles // 2))
    else:
        sel = (tile_of % stride == 0) | (rng.random(tiles)[tile_of] < 0.1)
    joints = np.where(rng.random(n) < 0.6)[0]
    if len(joints) == 0:
        joints = np.array([rng.integers(0, n)])
    eps = 0.0 if rng.random() < 0.15 else 10.0 ** rng.uniform(-12, 0)
    idx = np.where(sel)[0]
    dq[idx[:, None], joints[None, :]] *= eps
    ddq[idx[:, None], joints[None, :]] *= eps
    if rng.random() < 0.3:
        q[idx[:, None], joints[None, :]] = q[idx[0], joints][None, :]


@pytest.mark.parametrize("case", range(16))
def test_cholqr_route_on_randomly_doctored_batches(case):
    """Whatever path the device takes (tools/cholqr_fuzz.py prints it: of these 16 cases eleven end in round 0, four in round 1, one in
    the stand-by call), the factor is numpy's Householder factor: R'R = M'M to 1e-13, singular values to 1e-10."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    urdf, base, tool = CHOLQR_CASES[case % 2]
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P, N = ref.n, ref.P, 66000
    rng = np.random.default_rng(case)
    q, dq, ddq = trajectory_batch(case, N, n)
    _doctor(rng, q, dq, ddq, N, n)
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * rng.normal(size=(N, n))
    M = _oracle_rows(ref, q, dq, ddq, tau)
    R1 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))).cpu().numpy()
    G = M.T @ M
    assert np.allclose(np.tril(R1, -1), 0.0)
    assert np.abs(R1.T @ R1 - G).max() <= 1e-13 * np.abs(G).max()
    s_ref, s_gpu = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False), np.linalg.svd(R1, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-10
    assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])


@pytest.mark.parametrize("N", [6000, CHOLQR_N])
def test_cholqr_route_with_component_columns(N):
    """rdyn_identification_tsqr above the route's threshold: [Y | friction / spring columns | tau_meas] through the preconditioned
    CholeskyQR route (the component columns ride in the LDS tile as one more 16-column block): R'R = M'M, the singular values of
    numpy's Householder factor of the oracle's rows, the friction coefficients come back from the factor, accumulation."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.components import FRICTION1, FRICTION2, SPRING, ComponentSet
    from rosdyn_amd.gram import solve_r_factor
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    chain, ref = Chain(path, "base_link", "wrist_3_link", GRAV), OracleChain(path, "base_link", "wrist_3_link", GRAV)
    n, P = 6, 60
    q, dq, ddq = trajectory_batch(808, N, n)
    kinds = [FRICTION1, FRICTION2, SPRING]
    specs, dicts = [], []
    for j in range(n):
        ty = kinds[j % 3]
        par = [0.5 + 0.1 * j, 1.0 + 0.2 * j] + ([0.05] if ty == FRICTION2 else [])
        specs.append((ty, j, 1e-3, 5.0, par))
        dicts.append(dict(type=ty, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=par))
    comps = ComponentSet(dicts, n)
    K = comps.columns
    Cm, tau_c = components_regressor(specs, n, q, dq)
    rng = np.random.default_rng(11)
    tau = ref.joint_torque(q, dq, ddq) + tau_c + 1e-3 * rng.normal(size=(N, n))
    M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), Cm.reshape(-1, K), tau.reshape(-1)])
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    R1 = chain.getIdentificationTsqr(comps, *args).cpu().numpy()
    n1 = P + K + 1
    assert R1.shape == (n1, n1) and np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-11 * np.abs(G).max()
    s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9
    x, rank = solve_r_factor(R1, P + K, rtol=1e-9)
    truth = np.concatenate([sp[4] for sp in specs])
    assert np.abs(x[P:] - truth).max() < 0.01
    R2 = chain.getIdentificationTsqr(comps, *args, out=torch.from_numpy(R1).cuda(), accumulate=True).cpu().numpy()
    assert np.allclose(np.tril(R2, -1), 0.0) and np.abs(R2.T @ R2 - 2 * G).max() <= 1e-11 * np.abs(G).max()
