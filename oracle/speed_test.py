#!/usr/bin/env python3
"""BASELINE.json configs[0] ("rosdyn_speed_test 6-DOF URDF, 10 000 random samples on the CPU path"): the harness of
rosdyn_core/test/rosdyn_speed_test.cpp restated on the CPU oracle -- TEST INFRASTRUCTURE (plumbing, no GPU).
Prints the mean microseconds per call next to the numbers the reference publishes (README.md:29-45, laptop
Asus PU551J, unknown CPU).  Usage: python -m oracle.speed_test [urdf base tool]"""
import ctypes as C
import os
import sys

from .oracle import OracleChain, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
README_US = [0.75970, 1.06562, 1.25589, 1.25351, 1.51663, 1.83826, 2.68916, 3.76733, 10.06761, None]
NAMES = ["pose", "jacobian", "velocity twists for all links", "linear acceleration twists", "non linear acceleration twists",
         "acceleration twists for all links", "jerk twists for all links", "joint torque", "joint inertia",
         "regressor (not timed by the reference)"]


def run(urdf, base, tool, ntrial=10000, seed=0x5EED0001):
    c = OracleChain(urdf, base, tool, (0.0, 0.0, -9.806))
    f = lib().orc_speed_test
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    out = (C.c_double * 10)()
    f(c._h, ntrial, seed, out)
    return list(out)


def main():
    if len(sys.argv) >= 4:
        urdf, base, tool = sys.argv[1:4]
    else:
        urdf, base, tool = os.path.join(ROOT, "tests", "fixtures", "ur10_like.urdf"), "base_link", "tool0"
    us = run(urdf, base, tool)
    print("average on 10000 trials (CPU oracle, single thread); reference README.md column = Asus PU551J laptop")
    for name, t, r in zip(NAMES, us, README_US):
        print("computation time %-42s = %8.5f [us]   reference README: %s" % (name, t, "%8.5f" % r if r else "   n/a"))


if __name__ == "__main__":
    main()
