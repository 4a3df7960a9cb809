"""GPU tests of the round-4 factor routes (VERDICT r3 "Next round" item 1): the identification step's R factor of [Y | C | tau_meas]
on the reference's own chains -- 7 input joints with friction / spring columns (friction_polynomial1.h:126, ideal_spring.h:64 stacked
beside getRegressor, README.md:15), chains with fixed frames (ur10 base_link -> tool0, test.cpp:47-48; a Panda with flange and hand)
through the reduced companion -- the LDS-resident Householder folds that stand by for those shapes (rdyn_tsqr_wide.hip), and
rdyn_tsqr on materialised matrices of up to 112 columns with its matrix-core route.  Oracle: numpy.linalg.qr / the Gram matrix of the
C oracle's rows."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)


def _components(n, kinds):
    from rosdyn_amd.components import ComponentSet
    specs, dicts = [], []
    for j in range(n):
        ty = kinds[j % len(kinds)]
        par = [0.5 + 0.1 * j, 1.0 + 0.2 * j] + ([0.05] if ty == 1 else [])
        specs.append((ty, j, 1e-3, 5.0, par))
        dicts.append(dict(type=ty, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=par))
    return ComponentSet(dicts, n), specs


def _ident_case(urdf, base, tool, N, kinds, seed=5, noise=1e-3):
    """chain, oracle rows M = [Y | C | tau], torch inputs, component set, (P, K)"""
    import torch
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P = ref.n, ref.P
    q, dq, ddq = trajectory_batch(900 + seed + N % 1000, N, n)
    comps, specs = _components(n, kinds)
    K = comps.columns
    Cm, tau_c = components_regressor(specs, n, q, dq)
    rng = np.random.default_rng(seed)
    tau = ref.joint_torque(q, dq, ddq) + tau_c + noise * rng.normal(size=(N, n))
    M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), Cm.reshape(-1, K), tau.reshape(-1)])
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    return chain, M, args, comps, specs, P, K


def _check_factor(R1, M, tol_g=1e-11, tol_s=1e-9):
    n1 = M.shape[1]
    assert R1.shape == (n1, n1) and np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= tol_g * np.abs(G).max()
    if M.shape[0] >= n1:
        s_ref = np.linalg.svd(np.linalg.qr(M, mode="r"), compute_uv=False)
        s_gpu = np.linalg.svd(R1, compute_uv=False)
        keep = s_ref > 1e-9 * s_ref[0]
        assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= tol_s
        assert np.all(s_gpu[~keep] <= 1e-8 * s_ref[0])


FRICTION1, FRICTION2, SPRING = 0, 1, 2
IDENT_CHAINS = [
    ("panda_like.urdf", "link0", "link7", [FRICTION1]),                       # config 3's arm + 7 first-order friction components (K = 14)
    ("ur10_public.urdf", "base_link", "tool0", [FRICTION1, FRICTION2, SPRING]),  # the reference's benchmark chain (9 joints, 6 inputs) + 6 mixed
    ("panda_like.urdf", "link0", "hand", [FRICTION1]),                         # 9 joints, 7 inputs: reduced companion of 7 joints + components
    ("ur10_like.urdf", "base_link", "tool0", [FRICTION2]),                     # one fixed tail joint, K = 18
]
IDENT_IDS = ["panda7_f1", "ur10_public_tool0_mixed", "panda_hand_f1", "ur10_tool0_f2"]


def test_component_kinds_match_the_library():
    from rosdyn_amd import components as C
    assert (C.FRICTION1, C.FRICTION2, C.SPRING) == (FRICTION1, FRICTION2, SPRING)


@pytest.mark.parametrize("case", IDENT_CHAINS, ids=IDENT_IDS)
@pytest.mark.parametrize("N", [1, 33, 2000])
def test_identification_factor_small_batches(case, N):
    """Below 4 096 samples: the Householder folds -- in registers for swept chains of <= 6 joints, with the factor in LDS for the 7-joint
    arms (rdyn_tsqr_wide.hip).  R'R = M'M against the oracle's rows, numpy's singular values, both input layouts, accumulation, and the
    friction / spring coefficients come back from the factor."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.gram import solve_r_factor
    urdf, base, tool, kinds = case
    chain, M, args, comps, specs, P, K = _ident_case(urdf, base, tool, N, kinds)
    R1 = chain.getIdentificationTsqr(comps, *args).cpu().numpy()
    _check_factor(R1, M)
    R1e = chain.getIdentificationTsqr(comps, *(x.t().contiguous() for x in args), layout="element").cpu().numpy()
    assert np.array_equal(R1, R1e)
    assert np.array_equal(R1, chain.getIdentificationTsqr(comps, *args).cpu().numpy())      # reproducible
    if N == 2000:
        h = 777
        first = chain.getIdentificationTsqr(comps, *(x[:h].contiguous() for x in args))
        acc = chain.getIdentificationTsqr(comps, *(x[h:].contiguous() for x in args), out=first.clone(), accumulate=True).cpu().numpy()
        G = M.T @ M
        assert np.allclose(np.tril(acc, -1), 0.0) and np.abs(acc.T @ acc - G).max() <= 1e-11 * np.abs(G).max()
        x, rank = solve_r_factor(R1, P + K, rtol=1e-9)
        truth = np.concatenate([sp[4] for sp in specs])
        assert np.abs(x[P:] - truth).max() < 0.02


@pytest.mark.parametrize("case", IDENT_CHAINS, ids=IDENT_IDS)
@pytest.mark.parametrize("N", [4096, 66000])
def test_identification_factor_preconditioned_route(case, N):
    """From 4 096 samples on: preconditioned CholeskyQR on the matrix cores (two wave pairs on four SIMDs for the 7-joint arms with
    component columns), the reduced companion swept and the factor expanded with diag(E, I_K, 1) for the chains with fixed frames.
    The device's own acceptance (route 1, round 0) is read back."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import solve_r_factor
    import ctypes as C
    urdf, base, tool, kinds = case
    chain, M, args, comps, specs, P, K = _ident_case(urdf, base, tool, N, kinds)
    nbytes = lib().rdyn_identification_tsqr_workspace_bytes(chain._h, C.cast(comps._arr, C.c_void_p), comps.n_comps)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    R1 = chain.getIdentificationTsqr(comps, *args, workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws, components=comps)
    assert rep["route"] == 1 and rep["stage"] == 0 and 0 < rep["gamma"][0] <= 1e4 and 0.99 <= rep["rho"][0] <= 4.0, rep
    _check_factor(R1, M)
    x, rank = solve_r_factor(R1, P + K, rtol=1e-9)
    truth = np.concatenate([sp[4] for sp in specs])
    assert np.abs(x[P:] - truth).max() < 0.01
    G = M.T @ M
    R2 = chain.getIdentificationTsqr(comps, *args, out=torch.from_numpy(R1).cuda(), accumulate=True).cpu().numpy()
    assert np.allclose(np.tril(R2, -1), 0.0) and np.abs(R2.T @ R2 - 2 * G).max() <= 1e-11 * np.abs(G).max()
    assert np.array_equal(R1, chain.getIdentificationTsqr(comps, *args).cpu().numpy())


@pytest.mark.parametrize("kinds,K_expected", [([FRICTION2], 21), ([FRICTION2, FRICTION1, FRICTION1], 17), ([SPRING, FRICTION1], 14)])
def test_seven_joint_identification_factor_at_every_column_shift(kinds, K_expected):
    """k_regressor_pgram_solo is instantiated per quantised column shift (rdyn_pgram_solo.hip: 11 up to 14 component columns, 4 up to 21):
    second-order friction on every joint (K = 21, shift 4), a mix (K = 17: room for 8, quantised to 4), springs + friction (K = 14, shift
    11) -- R'R = M'M against the oracle's rows on the preconditioned route, accepted in round 0, reproducible bit for bit."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    import ctypes as C
    N = 20000
    chain, M, args, comps, specs, P, K = _ident_case("panda_like.urdf", "link0", "link7", N, kinds, seed=11)
    assert K == K_expected
    nbytes = lib().rdyn_identification_tsqr_workspace_bytes(chain._h, C.cast(comps._arr, C.c_void_p), comps.n_comps)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    R1 = chain.getIdentificationTsqr(comps, *args, workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws, components=comps)
    assert rep["route"] == 1 and rep["stage"] == 0, rep
    _check_factor(R1, M)
    assert np.array_equal(R1, chain.getIdentificationTsqr(comps, *args).cpu().numpy())


@pytest.mark.parametrize("tool,N", [("link3", 4096), ("link4", 5000), ("link2", 4500)])
def test_identification_factor_short_chains_above_the_threshold(tool, N):
    """chains of 2..4 joints with component columns and >= 4 096 samples took the preconditioned route, whose
    subsample kernel is only built for 5..7 joints (RDYN_ERR_HIP).  They keep the Householder folds."""
    pytest.importorskip("torch")
    chain, M, args, comps, specs, P, K = _ident_case("panda_like.urdf", "link0", tool, N, [FRICTION1, SPRING])
    R1 = chain.getIdentificationTsqr(comps, *args).cpu().numpy()
    _check_factor(R1, M)


def test_factor_of_a_chain_with_a_single_input_joint():
    """a chain whose reduced companion has ONE joint is swept as it is (the sweeping kernels start at two joints)."""
    torch = pytest.importorskip("torch")
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "ur10_public.urdf")
    chain, ref = Chain(path, "base_link", "shoulder_link", GRAV), OracleChain(path, "base_link", "shoulder_link", GRAV)
    assert ref.n == 1 and chain.getJointsNumber() >= 2
    N = 1500
    q, dq, ddq = trajectory_batch(31, N, 1)
    tau = ref.joint_torque(q, dq, ddq)
    M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, ref.P), tau.reshape(-1)])
    R1 = chain.getRegressorTsqr(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau))).cpu().numpy()
    _check_factor(R1, M)


def test_stand_by_of_the_seven_joint_identification_factor():
    """The subsample's tiles move 1e-9 times slower than the batch: neither round is accepted and the device starts the stand-by -- for a
    7-joint arm with component columns the LDS-resident Householder folds (the register-resident ones do not hold 86 columns)."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    import ctypes as C
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    path = os.path.join(FIXTURES, "panda_like.urdf")
    chain, ref = Chain(path, "link0", "link7", GRAV), OracleChain(path, "link0", "link7", GRAV)
    n, P, N = 7, 70, 40000
    q, dq, ddq = trajectory_batch(1234, N, n)
    tiles = (N + 15) // 16
    stride = max(1, tiles // 1024)
    stride += 1 if (stride > 1 and stride % 2 == 0) else 0
    sub = (np.arange(N) // 16) % stride == 0
    dq[sub] *= 1e-9
    ddq[sub] *= 1e-9
    comps, specs = _components(n, [FRICTION1])
    K = comps.columns
    Cm, tau_c = components_regressor(specs, n, q, dq)
    tau = ref.joint_torque(q, dq, ddq) + tau_c + 1e-3 * np.random.default_rng(2).normal(size=(N, n))
    M = np.column_stack([ref.regressor(q, dq, ddq).reshape(-1, P), Cm.reshape(-1, K), tau.reshape(-1)])
    nbytes = lib().rdyn_identification_tsqr_workspace_bytes(chain._h, C.cast(comps._arr, C.c_void_p), comps.n_comps)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    R1 = chain.getIdentificationTsqr(comps, *(torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)), workspace=ws).cpu().numpy()
    rep = chain.lastTsqrReport(N, ws, components=comps)
    assert rep["route"] == 1 and rep["stage"] >= 1, rep
    _check_factor(R1, M, tol_g=1e-12)
    # and with RDYN_TSQR_ROUTE-independent means: the plain Householder answer of a batch below the threshold agrees with numpy too
    # (that route is the stand-by's code)
    h = 3000
    Rh = chain.getIdentificationTsqr(comps, *(torch.from_numpy(x[:h].copy()).cuda() for x in (q, dq, ddq, tau))).cpu().numpy()
    _check_factor(Rh, M[: h * n])


@pytest.mark.parametrize("rows,n,with_b", [(1000, 70, True), (33, 100, False), (5000, 111, True), (200, 64, True), (129, 65, False), (9000, 95, True)])
def test_wide_matrix_householder_matches_numpy_qr(rows, n, with_b):
    """rdyn_tsqr beyond the 64 columns a wave's registers hold (and below the row count of the matrix-core route): LDS-resident folds."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.gram import tsqr
    rng = np.random.default_rng(rows + n)
    A = rng.normal(size=(rows, n)) * np.logspace(0, -3, n)[None, :]
    b = rng.normal(size=rows) if with_b else None
    At = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()
    R1 = tsqr(At, torch.from_numpy(b).cuda() if with_b else None).cpu().numpy()
    M = np.column_stack([A, b]) if with_b else A
    assert np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    assert np.abs(R1.T @ R1 - G).max() <= 1e-12 * np.abs(G).max()
    if rows >= M.shape[1]:
        Rq = np.linalg.qr(M, mode="r")
        s = np.sign(np.diag(R1)) * np.sign(np.diag(Rq))
        s[s == 0] = 1.0
        assert np.abs(R1 - s[:, None] * Rq).max() <= 1e-10 * np.abs(Rq).max()
    # accumulate: a second fold of the same rows
    R2 = tsqr(At, torch.from_numpy(b).cuda() if with_b else None, out=torch.from_numpy(R1).cuda(), accumulate=True).cpu().numpy()
    assert np.allclose(np.tril(R2, -1), 0.0) and np.abs(R2.T @ R2 - 2 * G).max() <= 1e-12 * np.abs(G).max()


@pytest.mark.parametrize("rows,n,with_b", [(40000, 30, True), (33000, 64, False), (70000, 90, True), (100000, 95, True), (50000, 16, False),
                                           (60000, 96, True), (70000, 104, True), (90000, 111, True), (40000, 112, False), (50000, 100, False)])
def test_matrix_tsqr_on_the_matrix_cores(rows, n, with_b):
    """rdyn_tsqr from 32 768 rows on, factors of <= 112 columns: the preconditioned CholeskyQR route fed from memory (k_pgram_rows; the
    dense steps keep their two squares in LDS up to 96 columns, in the workspace beyond).
    Columns spread over six orders of magnitude, two exactly dependent columns: the factor reproduces the Gram matrix and numpy's
    singular values; the device's acceptance is read back."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import tsqr, tsqr_last_report
    rng = np.random.default_rng(rows + n)
    A = rng.normal(size=(rows, n)) * np.logspace(0, -6, n)[None, :]
    if n >= 30:
        A[:, 7] = A[:, 3] - 2.0 * A[:, 5]       # structurally dependent columns, as a regressor has them
        A[:, 20] = 0.0
    b = A @ rng.normal(size=n) + 1e-3 * rng.normal(size=rows) if with_b else None
    n1 = n + (1 if with_b else 0)
    At = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()
    bt = torch.from_numpy(b).cuda() if with_b else None
    ws = torch.empty((lib().rdyn_tsqr_workspace_bytes(n1),), dtype=torch.uint8, device="cuda")
    R1 = tsqr(At, bt, workspace=ws).cpu().numpy()
    rep = tsqr_last_report(n1, rows, ws)
    assert rep["route"] == 1, rep
    M = np.column_stack([A, b]) if with_b else A
    assert np.allclose(np.tril(R1, -1), 0.0)
    G = M.T @ M
    d = np.sqrt(np.diag(G))
    d[d == 0] = 1.0
    assert np.abs((R1.T @ R1 - G) / np.outer(d, d)).max() <= 1e-11       # column-equilibrated: every column to its own scale
    s_ref = np.linalg.svd(np.linalg.qr(M / d, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1 / d, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9
    R2 = tsqr(At, bt, out=torch.from_numpy(R1).cuda(), accumulate=True).cpu().numpy()
    assert np.allclose(np.tril(R2, -1), 0.0) and np.abs((R2.T @ R2 - 2 * G) / np.outer(d, d)).max() <= 1e-11
    assert np.array_equal(R1, tsqr(At, bt).cpu().numpy())


def test_wide_factor_defers_nothing_on_a_well_conditioned_matrix():
    """85..96 columns: the flags of columns 84..95 used to share their workspace words with the diagnostics (gamma, rho), so those
    columns were reported -- and treated -- as deferred.  A well-conditioned matrix defers nothing and is accepted in round 0."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import tsqr, tsqr_last_report
    rng = np.random.default_rng(11)
    rows, n = 50000, 95
    A = rng.normal(size=(rows, n))
    b = rng.normal(size=rows)
    ws = torch.empty((lib().rdyn_tsqr_workspace_bytes(n + 1),), dtype=torch.uint8, device="cuda")
    R1 = tsqr(torch.from_numpy(np.ascontiguousarray(A.T)).cuda(), torch.from_numpy(b).cuda(), workspace=ws).cpu().numpy()
    rep = tsqr_last_report(n + 1, rows, ws)
    assert rep["route"] == 1 and rep["stage"] == 0 and rep["n_deferred"] == 0, rep
    assert 0.0 < rep["gamma"][0] < 100.0 and 0.0 < rep["rho"][0] <= 4.0, rep
    M = np.column_stack([A, b])
    G = M.T @ M
    assert np.allclose(np.tril(R1, -1), 0.0) and np.abs(R1.T @ R1 - G).max() <= 1e-12 * np.abs(G).max()


def test_matrix_tsqr_beyond_the_dense_kernels_width():
    """97..112 columns with many rows: the dense steps of the preconditioned route keep their squares in the workspace (round 5: until
    then the LDS-resident folds took these shapes) -- route 1, accepted in round 0 on a well-conditioned matrix, nothing deferred."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import tsqr, tsqr_last_report
    rng = np.random.default_rng(5)
    rows, n = 60000, 104
    A = rng.normal(size=(rows, n))
    b = rng.normal(size=rows)
    ws = torch.empty((lib().rdyn_tsqr_workspace_bytes(n + 1),), dtype=torch.uint8, device="cuda")
    R1 = tsqr(torch.from_numpy(np.ascontiguousarray(A.T)).cuda(), torch.from_numpy(b).cuda(), workspace=ws).cpu().numpy()
    rep = tsqr_last_report(n + 1, rows, ws)
    assert rep["route"] == 1 and rep["stage"] == 0 and rep["n_deferred"] == 0, rep
    assert 0.0 < rep["gamma"][0] < 100.0 and 0.0 < rep["rho"][0] <= 4.0, rep
    M = np.column_stack([A, b])
    G = M.T @ M
    assert np.allclose(np.tril(R1, -1), 0.0) and np.abs(R1.T @ R1 - G).max() <= 1e-12 * np.abs(G).max()


@pytest.mark.parametrize("seed", range(8))
def test_doctored_wide_matrices_on_the_matrix_cores(seed):
    """97..112 columns, doctored: column scales over eight orders of magnitude, exactly dependent and null columns, a subsample (every
    S-th 16-row group) that sees some columns at a vanishing scale -- whatever path the device takes (round 0, round 1, the stand-by),
    R'R = M'M to 1e-14 of the columns' own scales and numpy's singular values."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import tsqr, tsqr_last_report
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(97, 112))
    rows = int(rng.integers(140000, 200000))
    A = rng.normal(size=(rows, n)) * np.logspace(0, -8 * rng.random(), n)[rng.permutation(n)][None, :]
    dep = rng.choice(n, size=4, replace=False)
    A[:, dep[0]] = A[:, dep[1]] - 0.5 * A[:, dep[2]]
    if seed % 2:
        A[:, dep[3]] = 0.0
    # the rows the subsample sees: every gs-th 16-row group (rdyn_api.cpp: 8 192 groups, odd stride)
    groups = (rows + 15) // 16
    gs = max(1, groups // 8192)
    gs += 1 if (gs > 1 and gs % 2 == 0) else 0
    sub = (np.arange(rows) // 16) % gs == 0
    if seed >= 3:
        cols = rng.choice(n, size=int(rng.integers(1, 6)), replace=False)
        A[np.ix_(sub, cols)] *= 10.0 ** (-rng.integers(3, 12))
    b = A @ rng.normal(size=n) + 1e-3 * rng.normal(size=rows)
    ws = torch.empty((lib().rdyn_tsqr_workspace_bytes(n + 1),), dtype=torch.uint8, device="cuda")
    R1 = tsqr(torch.from_numpy(np.ascontiguousarray(A.T)).cuda(), torch.from_numpy(b).cuda(), workspace=ws).cpu().numpy()
    rep = tsqr_last_report(n + 1, rows, ws)
    assert rep["route"] == 1, rep
    M = np.column_stack([A, b])
    G = M.T @ M
    d = np.sqrt(np.diag(G))
    d[d == 0] = 1.0
    assert np.allclose(np.tril(R1, -1), 0.0)
    assert np.abs((R1.T @ R1 - G) / np.outer(d, d)).max() <= 1e-14 * n, rep
    s_ref = np.linalg.svd(np.linalg.qr(M / d, mode="r"), compute_uv=False)
    s_gpu = np.linalg.svd(R1 / d, compute_uv=False)
    keep = s_ref > 1e-9 * s_ref[0]
    assert np.abs(s_gpu[keep] / s_ref[keep] - 1.0).max() <= 1e-9, rep


def test_matrix_tsqr_solves_what_the_normal_equations_cannot_at_scale():
    """cond(A) = 1e9 with 80 columns and 65 536 rows (the matrix-core route): the solution comes back to ~cond * eps."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import lib
    from rosdyn_amd.gram import solve_r_factor, tsqr, tsqr_last_report
    rng = np.random.default_rng(8)
    m, n = 65536, 80
    U, _ = np.linalg.qr(rng.normal(size=(m, n)))
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    A = U @ np.diag(np.logspace(0, -9, n)) @ V.T
    x_true = V @ rng.normal(size=n)
    b = A @ x_true
    ws = torch.empty((lib().rdyn_tsqr_workspace_bytes(n + 1),), dtype=torch.uint8, device="cuda")
    R1 = tsqr(torch.from_numpy(np.ascontiguousarray(A.T)).cuda(), torch.from_numpy(b).cuda(), workspace=ws)
    rep = tsqr_last_report(n + 1, m, ws)
    assert rep["route"] == 1, rep
    x_qr, rank = solve_r_factor(R1, n, rtol=1e-13)
    assert rank == n
    err = np.abs(x_qr - x_true).max() / np.abs(x_true).max()
    assert err <= 1e-5, (err, rep)


@pytest.mark.parametrize("urdf,base,tool,N", [("panda_like.urdf", "link0", "hand", 4000000), ("ur10_public.urdf", "base_link", "tool0", 1000000),
                                              ("ur10_public_long.urdf", "base_link", "tcp", 1000000)], ids=["config3_size_panda_hand", "ur10_public_tool0", "14_joints"])
def test_full_size_identification_factor_reproduces_the_normal_equations(urdf, base, tool, N):
    """BASELINE configs[2] size with the friction columns of the identification step: two independent reductions of the same 28e6 rows
    of [Y | C | tau] -- the robust R factor (preconditioned CholeskyQR: k_regressor_pgram_solo for the 7-joint arm, reduced companion +
    expansion for the fixed frames) and the fp64-MFMA normal equations (rdyn_identification_gram) -- must agree: R1'R1 = [G c; c' bb].
    A size-independent property: the oracle cannot produce 28e6 rows in seconds."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd import Chain
    from rosdyn_amd.components import ComponentSet
    chain = Chain(os.path.join(FIXTURES, urdf), base, tool, GRAV)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    comps = ComponentSet([dict(type=j % 3, joint=j, min_velocity=1e-3, max_velocity=5.0, parameters=[0.3, 0.8, 0.02][:3 if j % 3 == 1 else 2]) for j in range(n)], n)
    K = comps.columns
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0003)
    q, dq, ddq, tau = (torch.rand((N, n), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(4))
    R1 = chain.getIdentificationTsqr(comps, q, dq, ddq, tau)
    G, c, bb = chain.getIdentificationGram(comps, q, dq, ddq, tau)
    C = P + K
    full = torch.zeros((C + 1, C + 1), dtype=torch.float64, device="cuda")
    full[:C, :C], full[:C, C], full[C, :C], full[C, C] = G, c, c, bb[0]
    # every column to its own scale -- down to 1e-7 of the largest: a structurally null column of this arm (m c_z of the second link: its
    # entries are rounding noise ~1e-17, which the robust factor reports as an exactly zero row) has no scale of its own
    d = full.diagonal().sqrt()
    d = d.clamp_min(1e-7 * d.max().item())
    err = ((R1.t() @ R1 - full) / torch.outer(d, d)).abs().max().item()
    assert err <= 1e-10, err
    assert torch.equal(torch.tril(R1, -1), torch.zeros_like(R1))
