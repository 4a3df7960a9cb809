#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files per (kernel, counter): tools/pmc_table.py <dir> [kernel substring]"""
import csv, glob, os, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub in k:
            acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-60s %-32s n=%d avg=%.4g" % (k, c, len(v), sum(v) / len(v)))
