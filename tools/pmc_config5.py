#!/usr/bin/env python3
"""HBM traffic per plan.run() of BASELINE configs[4] from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
tools/prof_r3_workloads.py config5 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE doubled on gfx950, KB units); written into
profiles/pmc_latest.json as `config5_stacked` (read by bench.py for extras.config5.roofline.traffic)."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
tot = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_image_sweep_multi" in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    tot[ctr] = {k: sum(v) / len(v) for k, v in per.items()}
    for k, v in tot[ctr].items():
        print("%-90s %s avg %.1f KB per launch" % (k[:90], ctr, v))
if tot["FETCH_SIZE"] and tot["WRITE_SIZE"]:
    traffic = 2.0 * 1024.0 * sum(tot["FETCH_SIZE"].values()) + 1024.0 * sum(tot["WRITE_SIZE"].values())
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_latest.json")
    j = json.load(open(path))
    j["config5_stacked"] = {"kernels": sorted(tot["WRITE_SIZE"]), "FETCH_SIZE_KB": sum(tot["FETCH_SIZE"].values()), "WRITE_SIZE_KB": sum(tot["WRITE_SIZE"].values()),
                            "traffic_bytes_per_launch": traffic, "correction": "traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 FETCH_SIZE halving); one plan.run() = both joint-count groups"}
    json.dump(j, open(path, "w"), indent=1, sort_keys=True)
    print("config5_stacked traffic per plan.run(): %.0f bytes" % traffic)
