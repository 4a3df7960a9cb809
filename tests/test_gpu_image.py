"""GPU tests of the LDS-staged one-thread-per-sample regressor kernel (rdyn_image.hip: per-sample drop-in images written as whole
128-byte lines, stacked matrix written as whole columns): every alignment class, ragged batches around the 64-sample wave, padded
image strides, a fixed tool frame behind the input joints (NA = NJ - 1), 7 input joints, both input layouts -- against the CPU
oracle and bit for bit against the element-major kernel, with poisoned outputs so that an unwritten byte is seen."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)
CASES = [("ur10_like.urdf", "base_link", "wrist_3_link"),   # NJ = NA = 6: 2 880-byte images, two alignment classes
         ("ur10_like.urdf", "base_link", "tool0"),           # NJ = 7, NA = 6 (fixed tool frame): 3 360-byte images
         ("panda_like.urdf", "link0", "link7"),              # NJ = NA = 7: 3 920-byte images, eight alignment classes
         ("panda_like.urdf", "link0", "link8"),              # NJ = 8, NA = 7
         # round 3: the reference's own chains in their public URDF form (rosdyn_speed_test.cpp:44-45, test.cpp:47-48)
         ("ur10_public.urdf", "base_link", "wrist_3_link"),  # fixed HEAD joint base_link -> base_link_inertia: NJ = 7, NA = 6
         ("ur10_public.urdf", "base_link", "flange"),        # fixed head + one fixed tail joint: NJ = 8
         ("ur10_public.urdf", "base_link", "tool0"),         # fixed head + two fixed tail joints: NJ = 9, P = 90
         ("panda_like.urdf", "link0", "hand"),               # two fixed tail joints behind 7 input joints: NJ = 9, P = 90
         ("panda_like.urdf", "link0", "hand", ["joint1", "joint2", "joint3", "joint4", "joint5", "joint6"]),  # joint7 not an input: 3 "fixed" tail joints
         ("ur10_public.urdf", "base_link", "tool0", ["shoulder_pan_joint", "shoulder_lift_joint", "elbow_joint", "wrist_1_joint", "wrist_2_joint"])]
IDS = ["6of6", "6of7", "7of7", "7of8", "h1_6of7", "h1_6of8", "h1_6of9", "7of9", "6of9_t3", "h1_5of9_t3"]


def _chain_and_ref(case):
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    urdf, base, tool = case[:3]
    path = os.path.join(FIXTURES, urdf)
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV, input_joint_names=case[3] if len(case) > 3 else None)
    if len(case) > 3:
        assert chain.setInputJointsName(case[3])
    return chain, ref


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_image_and_stacked_match_oracle_and_element_kernel(case, N):
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _chain_and_ref(case)
    n, P = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber()
    q, dq, ddq = trajectory_batch(900 + N, N, n)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    Yr, tr = ref.regressor(q, dq, ddq), ref.joint_torque(q, dq, ddq)
    nan = float("nan")
    Yp = torch.full((N + 1, P, n), nan, dtype=torch.float64, device="cuda")     # one image of slack: writes past the batch are seen
    Yp_v, tau_p = chain.getRegressor(tq, tdq, tddq, y_layout="per_sample", out=Yp[:N], with_torque=True)
    Ys = torch.full((P, N * n), nan, dtype=torch.float64, device="cuda")
    chain.getRegressor(tq, tdq, tddq, y_layout="stacked", out=Ys)
    Ye, tau_e = chain.getRegressor(*(x.t().contiguous() for x in (tq, tdq, tddq)), layout="element", with_torque=True)
    torch.cuda.synchronize()
    assert torch.isnan(Yp[N]).all()                                                # nothing written behind the last image
    a = Yp_v.cpu().numpy().transpose(0, 2, 1)
    b = Ys.cpu().numpy().reshape(P, N, n).transpose(1, 2, 0)
    c = Ye.cpu().numpy().transpose(2, 1, 0)
    assert not np.isnan(a).any() and not np.isnan(b).any()
    assert np.array_equal(a, c) and np.array_equal(b, c)                           # same arithmetic, three layouts
    assert np.array_equal(tau_p.cpu().numpy(), tau_e.cpu().numpy().T)
    assert np.abs(a - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())
    assert np.abs(tau_p.cpu().numpy() - tr).max() <= 1e-11 * max(1.0, np.abs(tr).max())


@pytest.mark.parametrize("pad", [2, 6, 16, 1], ids=["pad2", "pad6", "pad16", "pad1_rowpair_fallback"])
def test_padded_image_stride(pad):
    """stride_sample = n P + pad doubles: even pads keep 16-byte alignment (image kernel, all eight alignment classes), an odd pad
    falls back to the row-pair kernel.  The padding doubles between images must stay untouched."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _chain_and_ref(CASES[0])
    n, P, N = 6, 60, 333
    q, dq, ddq = trajectory_batch(31, N, n)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    ss = n * P + pad
    buf = torch.full((N * ss,), -7.0, dtype=torch.float64, device="cuda")
    b = Batch()
    b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, tq.data_ptr(), tdq.data_ptr(), tddq.data_ptr(), 0, 0
    b.stream = torch.cuda.current_stream().cuda_stream
    yl = RegressorLayout(ss, 1, n)
    check(lib().rdyn_regressor(chain._h, C.byref(b), None, buf.data_ptr(), C.byref(yl)))
    torch.cuda.synchronize()
    h = buf.cpu().numpy().reshape(N, ss)
    assert np.all(h[:, n * P:] == -7.0)                                            # padding untouched
    Yg = h[:, :n * P].reshape(N, P, n).transpose(0, 2, 1)
    Yr = ref.regressor(q, dq, ddq)
    assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())


@pytest.mark.parametrize("case", [CASES[0], CASES[6], CASES[7]], ids=["6of6", "h1_6of9", "7of9"])
def test_odd_double_offset_output_is_not_overrun(case):
    """a Y that is only 8-byte aligned (a view at an odd double offset, a C caller passing Y + 1) made the LDS-staged
    kernels' 16-byte copy-out write 8 bytes past every image.  Such calls now keep the row-pair kernel: the result is right and the
    doubles on either side of the output stay poisoned."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import Batch, RegressorLayout, check, lib
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _chain_and_ref(case)
    n, P, N = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber(), 130
    q, dq, ddq = trajectory_batch(77, N, n)
    tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))
    Yr = ref.regressor(q, dq, ddq)
    b = Batch()
    b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, tq.data_ptr(), tdq.data_ptr(), tddq.data_ptr(), 0, 0
    b.stream = torch.cuda.current_stream().cuda_stream
    for yl, unpack in ((RegressorLayout(n * P, 1, n), lambda h: h.reshape(N, P, n).transpose(0, 2, 1)),
                       (RegressorLayout(n, 1, N * n), lambda h: h.reshape(P, N, n).transpose(1, 2, 0))):
        for off in (1, 3):
            buf = torch.full((N * n * P + 8,), -7.0, dtype=torch.float64, device="cuda")
            check(lib().rdyn_regressor(chain._h, C.byref(b), None, buf.data_ptr() + 8 * off, C.byref(yl)))
            torch.cuda.synchronize()
            h = buf.cpu().numpy()
            assert np.all(h[:off] == -7.0) and np.all(h[off + N * n * P:] == -7.0)
            Yg = unpack(h[off:off + N * n * P])
            assert np.abs(Yg - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())


def test_image_kernel_with_prismatic_and_fixed_joints():
    """mixed_joints.urdf (revolute / prismatic / fixed, input joints need not be the first chain joints): whatever kernel the API
    picks, the per-sample image equals the oracle."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.samples import trajectory_batch
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    path = os.path.join(FIXTURES, "mixed_joints.urdf")
    base, tool = "world", "tip"
    chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
    n, N = chain.getActiveJointsNumber(), 257
    q, dq, ddq = trajectory_batch(5, N, n)
    Y = chain.getRegressor(*(torch.from_numpy(x).cuda() for x in (q, dq, ddq)), y_layout="per_sample")
    torch.cuda.synchronize()
    Yr = ref.regressor(q, dq, ddq)
    assert np.abs(Y.cpu().numpy().transpose(0, 2, 1) - Yr).max() <= 1e-11 * max(1.0, np.abs(Yr).max())


def test_regressor_strides_are_validated():
    """rdyn_regressor took any stride triple; a zero / negative stride or a sample stride that overflows the kernels'
    32-bit lane offsets wrote to wrong addresses.  They are RDYN_ERR_INVALID_ARGUMENT now (also per item of a multi-chain plan)."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd._lib import Batch, MultiItem, RegressorLayout, lib
    from rosdyn_amd import Chain
    chain = Chain(os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "wrist_3_link", GRAV)
    n, P, N = 6, 60, 8
    q = torch.zeros((N, n), dtype=torch.float64, device="cuda")
    Y = torch.zeros((N * n * P,), dtype=torch.float64, device="cuda")
    b = Batch()
    b.n_samples, b.q, b.dq, b.ddq, b.layout, b.device = N, q.data_ptr(), q.data_ptr(), q.data_ptr(), 0, 0
    for bad in (RegressorLayout(0, 1, n), RegressorLayout(n * P, -1, n), RegressorLayout(n * P, 1, 0), RegressorLayout(3000000, 1, n)):
        assert lib().rdyn_regressor(chain._h, C.byref(b), None, Y.data_ptr(), C.byref(bad)) == 1
        assert b"strides must be positive" in lib().rdyn_last_error()
    it = (MultiItem * 1)()
    it[0].chain, it[0].batch, it[0].Y, it[0].y_layout = chain._h, b, Y.data_ptr(), RegressorLayout(-1, N, n * N)
    h = C.c_void_p()
    assert lib().rdyn_multi_plan_create(C.cast(it, C.c_void_p), 1, C.byref(h)) == 1


def test_pick_output_buffer_returns_a_working_buffer():
    """rosdyn_amd.placement.pick_output_buffer: candidates are probed with the caller's launch, the fastest is kept, results are the same."""
    torch = pytest.importorskip("torch")
    from rosdyn_amd.placement import pick_output_buffer
    from rosdyn_amd.samples import trajectory_batch
    chain, ref = _chain_and_ref(CASES[0])
    n, P, N = chain.getActiveJointsNumber(), 10 * chain.getJointsNumber(), 20000
    q, dq, ddq = (torch.from_numpy(x).cuda() for x in trajectory_batch(5, N, n))
    tau = torch.empty((N, n), dtype=torch.float64, device="cuda")
    Y, info = pick_output_buffer(lambda Yc: chain.getRegressor(q, dq, ddq, y_layout="stacked", out=Yc, tau_out=tau), (P, N * n), torch.device("cuda", 0),
                                 max_candidates=5, batch=2)
    assert 2 <= info["candidates"] <= 5 and len(info["probe_ms"]) == info["candidates"] and 0 <= info["chosen"] < info["candidates"]
    assert info["probe_ms"][info["chosen"]] == min(info["probe_ms"])
    Y2 = torch.empty_like(Y)
    chain.getRegressor(q, dq, ddq, y_layout="stacked", out=Y2, tau_out=tau)
    chain.getRegressor(q, dq, ddq, y_layout="stacked", out=Y, tau_out=tau)
    assert torch.equal(Y, Y2)
