"""GPU tests of the fp64-MFMA Gram path (rdyn_gram / rdyn_regressor_gram).  Oracle: numpy A.T @ A on the CPU
oracle's regressor rows (extension row of SURVEY section 8c: no reference counterpart).
Tolerance: ||dG||_F <= 1e-10 ||G||_F (reduction-order dependent), BASELINE.md section 3."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    return torch


def _fro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("rows,P", [(16, 5), (1000, 15), (4099, 16), (12345, 60), (5000, 70), (3001, 80), (2000, 100)])
def test_gram_exact_integers(torch_cuda, rows, P):
    """Small-integer data: every product and partial sum is exactly representable, so G must be bit exact
    (catches any lane/row/column mapping error of the MFMA tiles, incl. asymmetric content and ragged tails)."""
    from rosdyn_amd.gram import gram
    torch = torch_cuda
    rng = np.random.default_rng(rows * 131 + P)
    A = rng.integers(-3, 4, size=(rows, P)).astype(np.float64)
    A[:, 0] += np.arange(rows) % 5            # asymmetric, column-dependent content
    b = rng.integers(-2, 3, size=rows).astype(np.float64)
    At = torch.from_numpy(np.ascontiguousarray(A.T)).cuda()   # (P, rows) = column-major rows x P
    G, c, bb = gram(At, torch.from_numpy(b).cuda())
    assert np.array_equal(G.cpu().numpy(), A.T @ A)
    assert np.array_equal(c.cpu().numpy(), A.T @ b)
    assert float(bb.item()) == float(b @ b)
    G2, c2, bb2 = gram(At, torch.from_numpy(b).cuda(), out=(G, c, bb), accumulate=True)
    assert np.array_equal(G2.cpu().numpy(), 2 * (A.T @ A))
    assert float(bb2.item()) == 2 * float(b @ b)


@pytest.mark.parametrize("urdf,base,tool,N,chunk", [("ur10_like.urdf", "base_link", "wrist_3_link", 5000, 2048),
                                                    ("panda_like.urdf", "link0", "link7", 3000, 0),
                                                    ("mixed_joints.urdf", "world", "tip", 2500, 1000),
                                                    # prismatic + revolute, every joint an input joint: the wave-pair kernel's generic
                                                    # (not all-revolute) unrolled sweeper
                                                    ("mixed_joints.urdf", "pedestal", "arm", 2000, 0)])
def test_regressor_gram_matches_oracle(torch_cuda, urdf, base, tool, N, chunk):
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import solve_base_parameters
    from rosdyn_amd.samples import trajectory_batch, uniform_pm1
    torch = torch_cuda
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, P = ref.n, ref.P
    q, dq, ddq = trajectory_batch(77, N, n)
    Y = ref.regressor(q, dq, ddq)                               # (N, n, P)
    pi = ref.nominal_parameters()
    tau_meas = Y @ pi + 1e-3 * uniform_pm1(99, (N, n))          # synthetic measurements
    A = Y.reshape(N * n, P)
    bvec = tau_meas.reshape(N * n)
    G_ref, c_ref, bb_ref = A.T @ A, A.T @ bvec, bvec @ bvec
    for layout in ("sample", "element"):
        if layout == "element":
            args = [torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq, tau_meas)]
        else:
            args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau_meas)]
        G, c, bb = chain.getRegressorGram(args[0], args[1], args[2], args[3], layout=layout, chunk_samples=chunk)
        assert _fro(G.cpu().numpy(), G_ref) <= 1e-10
        assert _fro(c.cpu().numpy(), c_ref) <= 1e-10
        assert abs(float(bb.item()) - bb_ref) <= 1e-10 * bb_ref
        Gh = G.cpu().numpy()
        assert np.array_equal(Gh, Gh.T)
    # the recovered parameters reproduce the measured torques (base-parameter check)
    x, rank = solve_base_parameters(G, c)
    assert rank < P                                              # structurally rank deficient
    assert np.abs(A @ x - bvec).max() <= 5e-3


def test_gram_of_materialised_regressor_equals_fused(torch_cuda):
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import gram
    from rosdyn_amd.samples import trajectory_batch
    torch = torch_cuda
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0", GRAV)
    n, P, N = 6, 70, 20000
    q, dq, ddq = (torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in trajectory_batch(3, N, n))
    Y, tau = chain.getRegressor(q, dq, ddq, layout="element", with_torque=True)     # (P, n, N)
    G1, c1, bb1 = gram(Y.reshape(P, n * N), tau.reshape(n * N))
    G2, c2, bb2 = chain.getRegressorGram(q, dq, ddq, tau, layout="element", chunk_samples=N)   # one chunk: same order
    assert torch.equal(G1, G2) and torch.equal(c1, c2) and torch.equal(bb1, bb2)
    G3, _, _ = chain.getRegressorGram(q, dq, ddq, tau, layout="element", chunk_samples=4096)
    assert float((G3 - G1).norm() / G1.norm()) <= 1e-12


def test_regressor_gram_with_permuted_subset_of_input_joints(torch_cuda):
    """Input joints not in chain order (setInputJointsName): the sorted view is swept (until round 4: the image path); same Gram."""
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    torch = torch_cuda
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    names = ["wrist_3_joint", "shoulder_pan_joint", "elbow_joint", "wrist_1_joint"]
    chain = Chain(path, "base_link", "tool0", GRAV)
    assert chain.setInputJointsName(names)
    ref = OracleChain(path, "base_link", "tool0", GRAV, names)
    N, n, P = 3000, ref.n, ref.P
    q, dq, ddq = trajectory_batch(17, N, n)
    Y = ref.regressor(q, dq, ddq)
    tau = ref.joint_torque(q, dq, ddq)
    A, bvec = Y.reshape(N * n, P), tau.reshape(N * n)
    args = [torch.from_numpy(np.ascontiguousarray(x.T)).cuda() for x in (q, dq, ddq, tau)]
    G, c, bb = chain.getRegressorGram(*args, layout="element")
    assert _fro(G.cpu().numpy(), A.T @ A) <= 1e-10
    assert _fro(c.cpu().numpy(), A.T @ bvec) <= 1e-10


@pytest.mark.parametrize("tool,names", [
    ("wrist_3_link", ["wrist_2_joint", "shoulder_pan_joint", "wrist_3_joint", "elbow_joint", "shoulder_lift_joint", "wrist_1_joint"]),
    ("tool0", ["wrist_3_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint", "shoulder_pan_joint"])])
@pytest.mark.parametrize("N", [700, 9000])
def test_normal_equations_and_r_factor_with_input_joints_in_any_order(torch_cuda, tool, names, N):
    """setInputJointsName in an order that is not the chain's (primitives_impl.h:705-737), every joint an input joint and with
    joints left out / fixed frames: the tile kernels sweep the sorted view and read q, Dq, DDq, tau_meas of every row through
    its index map (A'A, A'tau and R do not depend on the order of the rows inside a sample); friction columns follow their joint."""
    from oracle.oracle import OracleChain, components_regressor
    from rosdyn_amd import Chain
    from rosdyn_amd.components import ComponentSet
    from rosdyn_amd.samples import trajectory_batch
    torch = torch_cuda
    path = os.path.join(FIXTURES, "ur10_like.urdf")
    chain = Chain(path, "base_link", tool, GRAV)
    assert chain.setInputJointsName(names)
    ref = OracleChain(path, "base_link", tool, GRAV, names)
    n, P = ref.n, ref.P
    q, dq, ddq = trajectory_batch(23, N, n)
    rng = np.random.default_rng(N)
    tau = ref.joint_torque(q, dq, ddq) + 1e-3 * rng.normal(size=(N, n))
    A = ref.regressor(q, dq, ddq).reshape(N * n, P)
    M = np.column_stack([A, tau.reshape(-1)])
    Gr = M.T @ M
    args = [torch.from_numpy(x).cuda() for x in (q, dq, ddq, tau)]
    G, c, bb = chain.getRegressorGram(*args)
    full = np.zeros((P + 1, P + 1))
    full[:P, :P], full[:P, P], full[P, :P], full[P, P] = G.cpu().numpy(), c.cpu().numpy(), c.cpu().numpy(), float(bb.item())
    assert _fro(full, Gr) <= 1e-10
    R1 = chain.getRegressorTsqr(*args).cpu().numpy()
    assert np.allclose(np.tril(R1, -1), 0.0) and np.abs(R1.T @ R1 - Gr).max() <= 1e-11 * np.abs(Gr).max()
    # friction on input joints 0, 2 and n - 1 (INPUT indices: the joints named names[0], names[2], names[-1])
    specs = [(0, j, 1e-3, 5.0, [0.4 + 0.1 * j, 1.0]) for j in (0, 2, n - 1)]
    comps = ComponentSet([dict(type=0, joint=sp[1], min_velocity=1e-3, max_velocity=5.0, parameters=sp[4]) for sp in specs], n)
    Cm, tau_c = components_regressor(specs, n, q, dq)
    tau2 = tau + tau_c
    M2 = np.column_stack([A, Cm.reshape(N * n, comps.columns), tau2.reshape(-1)])
    G2r = M2.T @ M2
    args2 = args[:3] + [torch.from_numpy(tau2).cuda()]
    G2, c2, bb2 = chain.getIdentificationGram(comps, *args2)
    C = P + comps.columns
    full2 = np.zeros((C + 1, C + 1))
    full2[:C, :C], full2[:C, C], full2[C, :C], full2[C, C] = G2.cpu().numpy(), c2.cpu().numpy(), c2.cpu().numpy(), float(bb2.item())
    assert _fro(full2, G2r) <= 1e-10
    R3 = chain.getIdentificationTsqr(comps, *args2).cpu().numpy()
    assert np.allclose(np.tril(R3, -1), 0.0) and np.abs(R3.T @ R3 - G2r).max() <= 1e-11 * np.abs(G2r).max()


def test_config3_full_size_properties(torch_cuda):
    """BASELINE configs[2] size (Panda-like 7-DOF cut at link7: n = 7, P = 70, N = 4e6, measured torque = Y pi + noise):
    the Gram is additive over a split of the batch (second half accumulated onto the first), every path gives the same
    normal equations, the recovered base parameters predict the torques of an oracle-evaluated subset to the noise level,
    and bb = |tau_meas|^2."""
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.gram import r_factor, solve_base_parameters
    torch = torch_cuda
    path = os.path.join(FIXTURES, "panda_like.urdf")
    chain, ref = Chain(path, "link0", "link7", GRAV), OracleChain(path, "link0", "link7", GRAV)
    n, P, N = 7, 70, 4000000
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0003)
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3))
    tau = chain.getJointTorque(q, dq, ddq, layout="element")
    sigma = 1e-3
    tau_meas = tau + sigma * torch.randn((n, N), dtype=torch.float64, device="cuda", generator=gen)
    G, c, bb = chain.getRegressorGram(q, dq, ddq, tau_meas, layout="element")
    Gh, ch = G.cpu().numpy(), c.cpu().numpy()
    assert abs(bb.item() - float((tau_meas ** 2).sum())) <= 1e-10 * bb.item()
    assert np.array_equal(Gh, Gh.T)
    # additivity: halves, the second accumulated onto the first (element-major halves are strided views -> copy)
    h = N // 2
    first = [t[:, :h].contiguous() for t in (q, dq, ddq, tau_meas)]
    second = [t[:, h:].contiguous() for t in (q, dq, ddq, tau_meas)]
    out = chain.getRegressorGram(*first, layout="element")
    G2, c2, bb2 = chain.getRegressorGram(*second, layout="element", out=out, accumulate=True)
    assert _fro(G2.cpu().numpy(), Gh) <= 1e-11 and _fro(c2.cpu().numpy(), ch) <= 1e-11
    assert abs(bb2.item() - bb.item()) <= 1e-11 * bb.item()
    # the two-kernel path (regressor image through HBM in chunks) gives the same normal equations
    G3, c3, _ = chain.getRegressorGram(q, dq, ddq, tau_meas, layout="element", chunk_samples=262144)
    assert _fro(G3.cpu().numpy(), Gh) <= 1e-11 and _fro(c3.cpu().numpy(), ch) <= 1e-11
    # parameter recovery, checked on an oracle-evaluated subset
    x, rank = solve_base_parameters(G, c)
    R, perm, rank_r = r_factor(G, rtol=1e-10)
    assert rank == rank_r < P
    sub = slice(0, 2000)
    qs, dqs, ddqs = (t[:, sub].T.contiguous().cpu().numpy() for t in (q, dq, ddq))
    Ys = ref.regressor(qs, dqs, ddqs)
    pred, truth = Ys @ x, ref.joint_torque(qs, dqs, ddqs)
    assert np.abs(pred - truth).max() <= 20 * sigma / np.sqrt(N / 1000.0) + 1e-9 * np.abs(truth).max()


def test_identification_with_friction_columns_end_to_end(torch_cuda):
    """The identification step the reference's component regressors exist for (SURVEY 8f ranks 1 + 2): stack the rigid-body
    regressor and the friction columns [Y | C] in ONE element-major buffer, one MFMA Gram with the measured torque as b,
    host solve -- the friction coefficients come back, the rigid-body part predicts the torques."""
    from rosdyn_amd import Chain
    from rosdyn_amd.components import FRICTION1, ComponentSet
    from rosdyn_amd.gram import gram, solve_base_parameters
    torch = torch_cuda
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, 200000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    q, dq, ddq = (torch.rand((n, N), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(3))
    coulomb = [1.5, 2.0, 1.0, 0.4, 0.3, 0.2]
    viscous = [3.0, 2.5, 1.5, 0.5, 0.4, 0.1]
    comps = ComponentSet([dict(type=FRICTION1, joint=j, min_velocity=1e-4, max_velocity=10.0, parameters=[coulomb[j], viscous[j]])
                          for j in range(n)], n)
    K = comps.columns
    A = torch.empty((P + K, n, N), dtype=torch.float64, device="cuda")            # [Y | C], column-major (n N) x (P + K)
    tau = torch.empty((n, N), dtype=torch.float64, device="cuda")
    chain.getRegressor(q, dq, ddq, layout="element", out=A[:P], tau_out=tau)      # tau = Y pi
    comps.getRegressor(q, dq, layout="element", out=A[P:], tau_add=tau)           # tau += C phi
    sigma = 1e-2
    tau_meas = tau + sigma * torch.randn((n, N), dtype=torch.float64, device="cuda", generator=gen)
    G, c, bb = gram(A.reshape(P + K, n * N), tau_meas.reshape(n * N))
    x, rank = solve_base_parameters(G, c)
    assert P + K > rank >= K + 30                                                 # rigid-body part rank deficient, friction identifiable
    phi = x[P:]                                                                   # [coulomb_0, viscous_0, coulomb_1, ...]
    assert np.abs(phi[0::2] - np.array(coulomb)).max() < 20 * sigma / np.sqrt(N / 100.0)
    assert np.abs(phi[1::2] - np.array(viscous)).max() < 20 * sigma / np.sqrt(N / 100.0)
    pred = (A.reshape(P + K, n * N).T @ torch.from_numpy(x).cuda()).reshape(n, N)
    assert float((pred - tau).abs().max()) < 0.05                                 # noise-free torques reproduced
    # the same normal equations in ONE call, nothing materialised for the caller (rdyn_identification_gram); N is not a
    # multiple of the internal chunk, inputs in both layouts, and accumulated over two halves
    G1, c1, bb1 = chain.getIdentificationGram(comps, q, dq, ddq, tau_meas, layout="element")
    assert _fro(G1.cpu().numpy(), G.cpu().numpy()) <= 1e-12 and _fro(c1.cpu().numpy(), c.cpu().numpy()) <= 1e-12
    assert abs(bb1.item() - bb.item()) <= 1e-12 * bb.item()
    qs, dqs, ddqs, tms = (t.T.contiguous() for t in (q, dq, ddq, tau_meas))
    G2, c2, _ = chain.getIdentificationGram(comps, qs, dqs, ddqs, tms)
    assert _fro(G2.cpu().numpy(), G.cpu().numpy()) <= 1e-12 and _fro(c2.cpu().numpy(), c.cpu().numpy()) <= 1e-12
    h = 77777
    out = chain.getIdentificationGram(comps, qs[:h].contiguous(), dqs[:h].contiguous(), ddqs[:h].contiguous(), tms[:h].contiguous())
    G3, c3, bb3 = chain.getIdentificationGram(comps, qs[h:].contiguous(), dqs[h:].contiguous(), ddqs[h:].contiguous(), tms[h:].contiguous(),
                                              out=out, accumulate=True)
    assert _fro(G3.cpu().numpy(), G.cpu().numpy()) <= 1e-12 and abs(bb3.item() - bb.item()) <= 1e-12 * bb.item()
    # without components it is the regressor Gram
    G4, c4, _ = chain.getIdentificationGram(None, q, dq, ddq, tau_meas, layout="element")
    G5, c5, _ = chain.getRegressorGram(q, dq, ddq, tau_meas, layout="element")
    assert _fro(G4.cpu().numpy(), G5.cpu().numpy()) <= 1e-12 and _fro(c4.cpu().numpy(), c5.cpu().numpy()) <= 1e-12
