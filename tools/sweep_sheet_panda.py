#!/usr/bin/env python3
"""tools/sweep_sheet.py on the 7-joint arm of BASELINE.json configs[2] (panda_like link0 -> link7) and on the reference's own benchmark chain in
its public URDF form (ur10_public base_link -> tool0: 9 chain joints, 6 input joints)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_sheet import measure_sweeps  # noqa: E402

for fixture, base, tool in (("panda_like.urdf", "link0", "link7"), ("ur10_public.urdf", "base_link", "tool0")):
    r = measure_sweeps(fixture=fixture, base=base, tool=tool)
    print("sweep kernels, N = %d, %s (us per launch | GB/s algorithmic)" % (r["n_samples"], r["chain"]))
    for name in r["layouts"]["sample"]:
        s, e = r["layouts"]["sample"][name], r["layouts"]["element"][name]
        print("%-30s sample %8.1f %7.0f    element %8.1f %7.0f    %5.2f" % (name, s["ms"] * 1e3, s["GBps"], e["ms"] * 1e3, e["GBps"], s["ms"] / e["ms"]))
