#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel stats + PMC passes) of a bench.py run into a short text block."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-70s calls=%s total_ns=%s avg_ns=%s min=%s max=%s pct=%s" % (
        r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))

# per-dispatch durations of our kernels from the kernel trace
dur = defaultdict(list)
meta = {}
for r in rows("trace/**/*kernel_trace.csv"):
    name = r.get("Kernel_Name", "")
    if "k_local_sweep" in name or "k_base_sweep" in name or "k_gram" in name:
        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[name] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for k, v in dur.items():
    v.sort()
    print("dispatches %-60s n=%d avg_us=%.2f median_us=%.2f min_us=%.2f  vgpr/agpr/sgpr/lds/scratch/grid/wg=%s" % (
        k[:60], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, meta[k]))

for tag in ("fetch", "write"):
    acc = defaultdict(list)
    for r in rows("pmc_%s/**/*counter_collection.csv" % tag):
        name = r.get("Kernel_Name", "")
        if "k_local_sweep" in name or "k_base_sweep" in name or "k_gram" in name:
            acc[(name[:60], r.get("Counter_Name"))].append(float(r.get("Counter_Value", 0)))
    for (k, c), v in acc.items():
        print("pmc %-60s %s: n=%d avg=%.1f (KB units -> %.1f MB per launch)" % (k, c, len(v), sum(v) / len(v), sum(v) / len(v) / 1024.0))
