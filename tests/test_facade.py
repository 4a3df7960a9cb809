"""The C++ facade (rosdyn_amd/csrc/rosdyn_chain_facade.hpp) compiles as plain host C++ and the harness binary
links against the C-ABI; on a GPU box the harness runs (single-sample and batched calls)."""
import os
import subprocess

import pytest

from conftest import FIXTURES, ROOT

BIN = os.path.join(ROOT, "rosdyn_amd", "rdyn_speed_test")


def test_facade_header_compiles_standalone(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "rosdyn_chain_facade.hpp"\nint main() { return sizeof(rosdyn::Chain) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "rosdyn_amd", "csrc"), str(src)])


def test_harness_binary_is_built_and_reports_usage():
    assert os.path.exists(BIN), "run __graft_entry__.build()"
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_harness_runs_on_gpu():
    r = subprocess.run([BIN, os.path.join(FIXTURES, "ur10_like.urdf"), "base_link", "tool0", "200"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "joint torque + regressor" in r.stdout and "computation time regressor" in r.stdout


@pytest.mark.gpu
def test_facade_reports_reference_errors():
    r = subprocess.run([BIN, os.path.join(FIXTURES, "ur10_like.urdf"), "nope", "tool0", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "Base link not found" in r.stderr


@pytest.mark.gpu
def test_facade_single_sample_values_match_oracle():
    """The C++ facade's one-sample getters (host -> device -> kernel -> host) against the CPU oracle."""
    import numpy as np
    from oracle.oracle import OracleChain
    urdf = os.path.join(FIXTURES, "ur10_like.urdf")
    r = subprocess.run([BIN, urdf, "base_link", "tool0", "1", "dump"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vals = {l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in r.stdout.splitlines() if l and l[0] in "tYMTJILW"}
    ref = OracleChain(urdf, "base_link", "tool0", (0.0, 0.0, -9.806))
    n = ref.n
    q = np.array([[0.1 * (i + 1) for i in range(n)]])
    dq = np.array([[-0.05 * (i + 1) for i in range(n)]])
    ddq = np.array([[0.3 - 0.02 * i for i in range(n)]])

    def close(a, b):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())
    close(vals["tau"], ref.joint_torque(q, dq, ddq)[0])
    close(vals["Y"], ref.regressor(q, dq, ddq)[0].T.reshape(-1))          # column-major n x P
    close(vals["M"], ref.joint_inertia(q)[0].T.reshape(-1))
    close(vals["T"], ref.fk(q)[0, -1].T.reshape(-1))                       # column-major 3 x 4
    close(vals["J"], ref.jacobian(q)[0].T.reshape(-1))                     # column-major 6 x n
    close(vals["L"], ref.jacobian_link(q, ref.L // 2)[0].T.reshape(-1))    # getJacobianLink of the middle link
    # getWrench with external loads (base-link record), jerk parts summing to the jerk (tool link)
    dddq = np.array([[0.7 - 0.1 * i for i in range(n)]])
    ext = np.array([[[0.5 * (l + 1) - 0.3 * i for i in range(6)] for l in range(ref.L)]])
    _, w = ref.joint_torque(q, dq, ddq, ext=ext, wrenches=True)
    close(vals["W"][:6], w[0, 0])
    jerk = ref.ddtwist(q, dq, ddq, dddq)[0, -1]
    close(vals["W"][6:12], jerk)
    close(vals["W"][12:18], jerk)
    # computeLocalIk as in rosdyn_core/README.md:78-84: back to q from a displaced seed
    seed = q + np.array([[0.2 if i % 2 == 0 else -0.15 for i in range(n)]])
    rsol, rst, _ = ref.local_ik(ref.fk(q)[:, -1], seed, toll=1e-8, max_iter=50)
    assert vals["IK"][0] == 1.0 and rst[0] == 1
    assert np.abs(vals["IK"][1:] - rsol[0]).max() < 1e-9
