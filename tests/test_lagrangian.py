"""The oracle against physics in 3-D (VERDICT r3 "Next round" item 5; CPU only).  tools/lagrangian_check.py derives tau, the joint
inertia matrix and every regressor column (column p = d tau / d pi_p) from the Lagrangian of the parsed URDF by fp64 automatic
differentiation -- own URDF reader, own forward kinematics, no velocity recursion, no Newton-Euler -- and oracle/rosdyn_oracle.c (the
restatement of primitives_impl.h:1231-1272, 1321-1352, 1357-1379, 399-417) must agree.  The reference holds no numbers for this path
(test.cpp: no assertions) and cannot be built here: this is the pin that exists.  Revolute, prismatic and fixed joints, fixed head and
tail frames, links without inertial data."""
import pytest

from tools.lagrangian_check import CASES, compare


@pytest.mark.parametrize("urdf,base,tool", CASES, ids=["ur10_public_tool0", "panda_hand", "mixed_world_tip"])
def test_oracle_equals_euler_lagrange(urdf, base, tool):
    pytest.importorskip("torch")
    worst = compare(urdf, base, tool, n_samples=3)
    # observed 2e-16 .. 1e-15 (the two sides share nothing but the URDF file); the verdict's bar is 1e-9
    assert worst["tau"] <= 1e-12 and worst["M"] <= 1e-12 and worst["Y"] <= 1e-12, worst
    assert worst["pi"] <= 1e-14, worst                  # getNominalParameters = [m, m c, inertia about the link ORIGIN]
    assert worst["tau_from_pi"] <= 1e-12, worst         # the Lagrangian's own regressor times the physical parameters is its torque
