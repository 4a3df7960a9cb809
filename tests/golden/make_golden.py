#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- inputs and expected outputs for the hot path.

The expected values come from the INDEPENDENT numpy restatement (oracle/np_restatement.py), not from
the C oracle and not from the product, so the fixtures pin both.  (The reference itself cannot be
built or imported in this image -- no Eigen/ROS/urdfdom -- hence "parity unpinned" at that level;
see oracle/rosdyn_oracle.c.)  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.np_restatement import NpChain          # noqa: E402
from rosdyn_amd.samples import trajectory_batch    # noqa: E402

FIX = os.path.join(ROOT, "tests", "fixtures")
GRAV = (0.0, 0.0, -9.806)   # rosdyn_speed_test.cpp:61-62

# name -> (urdf, base, tool, gravity, input joint names or None, seed)
CASES = {
    "ur10_tool0":      ("ur10_like.urdf", "base_link", "tool0", GRAV, None, 0x5EED0001),
    "ur10_wrist3":     ("ur10_like.urdf", "base_link", "wrist_3_link", GRAV, None, 0x5EED0002),
    "panda_hand":      ("panda_like.urdf", "link0", "hand", GRAV, None, 0x5EED0003),
    "panda_link7":     ("panda_like.urdf", "link0", "link7", (0.3, -0.2, -9.7), None, 0x5EED0013),
    "mixed_world_tip": ("mixed_joints.urdf", "world", "tip", GRAV, None, 0x5EED0004),
    "mixed_sub_nograv": ("mixed_joints.urdf", "pedestal", "plate", (0.0, 0.0, 0.0), None, 0x5EED0014),
    "ur10_permuted":   ("ur10_like.urdf", "base_link", "tool0", GRAV,
                        ["wrist_3_joint", "shoulder_pan_joint", "elbow_joint", "wrist_1_joint"], 0x5EED0024),
    "planar_2r":       ("planar_2r.urdf", "base", "l2", GRAV, None, 0x5EED0034),
    # round 3: the reference's own chains in the topology of the public ur_description (test.cpp:47-48, rosdyn_speed_test.cpp:44-45):
    # a fixed joint in front of the six revolute ones, flange and tool0 behind them
    "ur10_public_tool0":  ("ur10_public.urdf", "base_link", "tool0", GRAV, None, 0x5EED0044),
    "ur10_public_flange": ("ur10_public.urdf", "base_link", "flange", (0.1, 0.2, -9.7), None, 0x5EED0054),
}
N = 16


def main():
    only = set(sys.argv[1:])   # names to (re)generate; none = all
    for name, (urdf, base, tool, g, inputs, seed) in CASES.items():
        if only and name not in only:
            continue
        c = NpChain(os.path.join(FIX, urdf), base, tool, g, inputs)
        q, dq, ddq = trajectory_batch(seed, N, c.n)
        # a few special rows: zero state, zero velocity, large angles (argument reduction of sin/cos)
        q[0] = 0; dq[0] = 0; ddq[0] = 0
        dq[1] = 0
        q[2] *= 40.0
        out = dict(urdf=urdf, base=base, tool=tool, gravity=np.array(g), seed=seed,
                   inputs=np.array(inputs if inputs else [], dtype="U64"),
                   q=q, dq=dq, ddq=ddq,
                   T=c.fk(q), J=c.jacobian(q), twist=c.twist(q, dq), dtwist=c.dtwist(q, dq, ddq),
                   tau=c.joint_torque(q, dq, ddq), Y=c.regressor(q, dq, ddq), M=c.joint_inertia(q),
                   pi=c.nominal_parameters())
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
        print(name, "n=%d nJ=%d" % (c.n, c.nJ))


if __name__ == "__main__":
    main()
