#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_mix.py file.s <substring of mangled name>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = None
for i, l in enumerate(txt):
    if re.match(r"^_Z\S*:", l) and pat in l:
        start = i
        break
assert start is not None, "kernel not found"
c = collections.Counter()
for l in txt[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end"):
        break
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    c[s.split()[0]] += 1
g = collections.Counter()
for op, n in c.items():
    if op.startswith("v_") and "f64" in op:
        g["valu_f64"] += n
    elif op.startswith("v_"):
        g["valu_other"] += n
    elif op.startswith("s_load"):
        g["s_load"] += n
    elif op.startswith("s_waitcnt"):
        g["s_waitcnt"] += n
    elif op.startswith("s_"):
        g["salu"] += n
    elif op.startswith("global_store"):
        g["gstore"] += n
    elif op.startswith("global_load"):
        g["gload"] += n
    elif op.startswith("scratch"):
        g["scratch"] += n
    elif op.startswith("ds_"):
        g["lds"] += n
    else:
        g[op] += n
print("total", sum(c.values()), dict(g))
print(c.most_common(30))
