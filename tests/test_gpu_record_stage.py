"""GPU tests (-m gpu) of the staged copy-out of the SAMPLE-MAJOR records (rosdyn_amd/csrc/rdyn_record_stage.h): the kinematic, torque
and inertia kernels write the drop-in layout (a sample's record contiguous: the memory image of the VectorOfAffine3d / VectorOfVector6d /
Matrix6Xd / MatrixXd / VectorXd objects of primitives_impl.h:884-912, 927-949, 981-1013, 1264-1293, 1357-1379) through wave-private LDS
in whole 128-byte lines.  The arithmetic is the element-major kernels' own, so the two layouts must agree BIT FOR BIT; the oracle
comparison proper is tests/test_gpu_parity.py (N = 2 000 there: 31 staged waves + a ragged one).

Covered: chains of 1 ... 10 and 14 joints (every instantiation's ring geometry: records of 96 (NJ + 1) / 48 (NJ + 1) bytes put the
samples of a wave into up to eight alignment classes), full waves + a ragged last wave, an output that does not start on a line (the
host then keeps the 8-byte stores), and guard bands around every output (a copy-out that wrote one chunk too many would show)."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu
GRAV = (0.0, 0.0, -9.806)

# (urdf, base, tool): 1 ... 10 chain joints; fixed joints at the head / tail / middle; prismatic joints (mixed_joints); 14: the
# run-time-length kernels of rdyn_long_kin.hip
CHAINS = [("ur10_like.urdf", "base_link", "shoulder_link"), ("ur10_like.urdf", "base_link", "upper_arm_link"),
          ("ur10_like.urdf", "base_link", "forearm_link"), ("ur10_like.urdf", "base_link", "wrist_1_link"),
          ("ur10_like.urdf", "base_link", "wrist_2_link"), ("ur10_like.urdf", "base_link", "wrist_3_link"),
          ("ur10_like.urdf", "base_link", "tool0"), ("mixed_joints.urdf", "world", "tip"), ("panda_like.urdf", "link0", "hand"),
          ("ur10_public.urdf", "base_link", "tool0"), ("ur10_public_long.urdf", "base_link", "tool0"),
          ("ur10_public_long.urdf", "base_link", "tcp")]


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _guarded(torch, shape, offset_doubles=0):
    """A contiguous float64 CUDA tensor of `shape` inside a larger NaN-filled buffer: (view, check()) -- check() asserts the guard bands
    (256 doubles in front of and behind the view) still hold their pattern.  offset_doubles shifts the view off its 256-byte alignment."""
    n = int(np.prod(shape))
    G = 256
    buf = torch.full((n + 2 * G + 32,), float("nan"), dtype=torch.float64, device="cuda")
    lo = G + offset_doubles
    view = buf[lo:lo + n].view(*shape)

    def check(what):
        assert bool(torch.isnan(buf[:lo]).all()) and bool(torch.isnan(buf[lo + n:]).all()), "%s: wrote outside its output" % what
        assert not bool(torch.isnan(view).any()), "%s: left part of its output unwritten" % what
    return view, check


def _calls(chain, q, dq, ddq, dddq, ext, link_name):
    """name -> (record shape, call(layout, inputs, out))"""
    n, L = chain.getActiveJointsNumber(), chain.getLinksNumber()
    import ctypes as C
    from rosdyn_amd._lib import check, lib

    def parts(which):
        def run(lay, x, out):
            b, N, _ = chain._batch(lay, x["q"], x["dq"], x["ddq"])
            p = [None, None, None]
            p[which] = out.data_ptr()
            check(lib().rdyn_twist_parts(chain._h, C.byref(b), x["dddq"].data_ptr(), *p))
        return run

    def jerk(which):
        def run(lay, x, out):
            b, N, _ = chain._batch(lay, x["q"], x["dq"], x["ddq"])
            p = [None, None]
            p[which] = out.data_ptr()
            check(lib().rdyn_jerk_parts(chain._h, C.byref(b), x["dddq"].data_ptr(), *p))
        return run

    return {
        "getTransformation": ((4, 3), lambda lay, x, out: chain.getTransformation(x["q"], layout=lay, out=out)),
        "getTransformations": ((L, 4, 3), lambda lay, x, out: chain.getTransformations(x["q"], layout=lay, out=out)),
        "getJacobian": ((n, 6), lambda lay, x, out: chain.getJacobian(x["q"], layout=lay, out=out)),
        "getJacobianLink": ((n, 6), lambda lay, x, out: chain.getJacobianLink(x["q"], link_name, layout=lay, out=out)),
        "getTwist": ((L, 6), lambda lay, x, out: chain.getTwist(x["q"], x["dq"], layout=lay, out=out)),
        "getDTwist": ((L, 6), lambda lay, x, out: chain.getDTwist(x["q"], x["dq"], x["ddq"], layout=lay, out=out)),
        "getDTwistLinearPart": ((L, 6), parts(0)),
        "getDTwistNonLinearPart": ((L, 6), parts(1)),
        "getDDTwist": ((L, 6), parts(2)),
        "getDDTwistLinearPart": ((L, 6), jerk(0)),
        "getDDTwistNonLinearPart": ((L, 6), jerk(1)),
        "getWrench": ((L, 6), lambda lay, x, out: chain.getWrench(x["q"], x["dq"], x["ddq"], x["ext"], layout=lay, out=out)),
        "getJointTorque": ((n,), lambda lay, x, out: chain.getJointTorque(x["q"], x["dq"], x["ddq"], layout=lay, out=out)),
        "getJointTorqueNonLinearPart": ((n,), lambda lay, x, out: chain.getJointTorqueNonLinearPart(x["q"], x["dq"], layout=lay, out=out)),
        "getJointInertia": ((n, n), lambda lay, x, out: chain.getJointInertia(x["q"], layout=lay, out=out)),
    }


@pytest.mark.parametrize("urdf,base,tool", CHAINS)
def test_sample_major_records_equal_element_major_bitwise(torch_cuda, urdf, base, tool):
    torch = torch_cuda
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import uniform_pm1
    chain = Chain(os.path.join(FIXTURES, urdf), base, tool, GRAV)
    n, L = chain.getActiveJointsNumber(), chain.getLinksNumber()
    long_many_inputs = chain.getJointsNumber() > 10 and n > 10
    N = 64 * 9 + 23  # nine staged waves and a ragged one
    host = uniform_pm1(0x5EED0600 + L, (4, N, n))
    ext_h = uniform_pm1(0x5EED0700 + L, (N, L, 6))
    xs = {k: torch.from_numpy(np.ascontiguousarray(host[i])).cuda() for i, k in enumerate(("q", "dq", "ddq", "dddq"))}
    xs["ext"] = torch.from_numpy(ext_h).cuda()
    xe = {k: v.T.contiguous() for k, v in xs.items() if k != "ext"}
    xe["ext"] = xs["ext"].permute(1, 2, 0).contiguous()
    link_name = chain.getLinksName()[max(1, L // 2)]
    for name, (rec, call) in _calls(chain, None, None, None, None, None, link_name).items():
        if long_many_inputs and name in ("getJointInertia",):
            continue
        ref = torch.empty(rec + (N,), dtype=torch.float64, device="cuda")
        call("element", xe, ref)
        want = ref.permute(len(rec), *range(len(rec))).contiguous()  # (N,) + rec
        for off in (0, 1):  # line-aligned output: the staged copy-out; one double off a line: the 8-byte stores
            out, check = _guarded(torch, (N,) + rec, off)
            call("sample", xs, out)
            torch.cuda.synchronize()
            check("%s (offset %d)" % (name, off))
            assert torch.equal(out, want), "%s: sample-major differs from element-major (offset %d, max |d| %.3e)" % (
                name, off, float((out - want).abs().max()))


def test_staged_records_match_oracle_full_waves(torch_cuda):
    """N = 4 096 (64 full waves, no ragged tail: every sample goes through the staged copy-out) against the C oracle."""
    torch = torch_cuda
    from oracle.oracle import OracleChain
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import trajectory_batch
    for urdf, base, tool in (("ur10_like.urdf", "base_link", "wrist_3_link"), ("panda_like.urdf", "link0", "hand")):
        path = os.path.join(FIXTURES, urdf)
        chain, ref = Chain(path, base, tool, GRAV), OracleChain(path, base, tool, GRAV)
        N = 4096
        q, dq, ddq = trajectory_batch(0x5EED0610, N, ref.n)
        tq, tdq, tddq = (torch.from_numpy(x).cuda() for x in (q, dq, ddq))

        def close(a, b, what):
            a = a.cpu().numpy()
            err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
            assert err <= 1e-11, (what, err)
        close(chain.getTransformations(tq).transpose(-1, -2), ref.fk(q), "T")
        close(chain.getTransformation(tq).transpose(-1, -2), ref.fk(q)[:, -1], "T_bt")
        close(chain.getJacobian(tq).transpose(-1, -2), ref.jacobian(q), "J")
        close(chain.getTwist(tq, tdq), ref.twist(q, dq), "twist")
        close(chain.getDTwist(tq, tdq, tddq), ref.dtwist(q, dq, ddq), "dtwist")
        close(chain.getJointTorque(tq, tdq, tddq), ref.joint_torque(q, dq, ddq), "tau")
        close(chain.getJointInertia(tq).transpose(-1, -2), ref.joint_inertia(q), "M")


@pytest.mark.parametrize("N", [1, 63, 64, 65, 127, 128, 129, 191, 320])
def test_batch_sizes_around_the_wave(torch_cuda, N):
    """Batches of less than a wave, exactly one wave, one sample more ... : every sample-major getter equals its element-major twin bit for
    bit (6- and 9-joint chains; the ragged wave keeps the 8-byte stores, full waves go through the staged copy-out)."""
    torch = torch_cuda
    from rosdyn_amd import Chain
    from rosdyn_amd.samples import uniform_pm1
    for urdf, base, tool in (("ur10_like.urdf", "base_link", "wrist_3_link"), ("ur10_public.urdf", "base_link", "tool0")):
        chain = Chain(os.path.join(FIXTURES, urdf), base, tool, GRAV)
        n, L = chain.getActiveJointsNumber(), chain.getLinksNumber()
        host = uniform_pm1(0x5EED0800 + N, (4, N, n))
        xs = {k: torch.from_numpy(np.ascontiguousarray(host[i])).cuda() for i, k in enumerate(("q", "dq", "ddq", "dddq"))}
        xs["ext"] = torch.from_numpy(uniform_pm1(0x5EED0900 + N, (N, L, 6))).cuda()
        xe = {k: v.T.contiguous() for k, v in xs.items() if k != "ext"}
        xe["ext"] = xs["ext"].permute(1, 2, 0).contiguous()
        for name, (rec, call) in _calls(chain, None, None, None, None, None, chain.getLinksName()[2]).items():
            ref = torch.empty(rec + (N,), dtype=torch.float64, device="cuda")
            call("element", xe, ref)
            want = ref.permute(len(rec), *range(len(rec))).contiguous()
            out, check = _guarded(torch, (N,) + rec, 0)
            call("sample", xs, out)
            torch.cuda.synchronize()
            check("%s N = %d" % (name, N))
            assert torch.equal(out, want), "%s N = %d: sample-major differs from element-major" % (name, N)
