#!/usr/bin/env python3
"""profiles/<round>/matrix_pipe_pmc.txt from the *_summary.txt files of tools/gpu_profile.sh round <round>: per MFMA kernel the matrix-pipe
utilisation SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1 024 SIMDs), instruction counts and the share of wave cycles spent waiting.
usage: tools/matrix_pipe_table.py profiles/r5"""
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r5"
rows = []
for w in ("gram", "cholqr", "ident", "tsqr_rows"):
    p = os.path.join(d, w + "_summary.txt")
    if not os.path.exists(p):
        continue
    c = {}
    for line in open(p):
        m = re.match(r"(.*?)\s+(SQ_\w+|GRBM_GUI_ACTIVE)\s+n=(\d+)\s+avg=([\d.e+]+)", line)
        if m:
            c.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    for k, v in c.items():
        if v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) <= 0 or "GRBM_GUI_ACTIVE" not in v:
            continue
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
        wait = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0
        name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        rows.append("%-10s %-46s MFMA busy %5.1f %%   MFMA instr %.3g   VALU instr %.3g   WAIT_INST_ANY / WAVE_CYCLES %4.0f %%"
                    % (w, name[:46], 100 * busy, v["SQ_INSTS_VALU_MFMA_MOPS_F64"], v.get("SQ_INSTS_VALU", 0), 100 * wait))
print("\n".join(rows))
