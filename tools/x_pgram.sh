#!/bin/bash
# tools/x_pgram.sh what lib...: the duration of pass B (k_regressor_pgram, the longest dispatch) under each library, from a kernel trace
# (timing experiments whose numbers are wrong end up on the stand-by route: the whole-call time of kbench says nothing about them)
WHAT=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for L in "$@"; do
  D=gpurun_out/x_pgram/$(basename $L .so)
  rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --output-format csv -d $D -- tools/_build/kbench $WHAT 3 $L > $D/log.txt 2>&1
  python3 - "$D" "$L" <<'PY'
import csv, glob, sys
best = {}
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if "pgram" in n or "gram_duo" in n:
            k = n[:70]
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            best.setdefault(k, []).append(d)
for k, v in best.items():
    v.sort()
    print('%-40s %-70s n=%d max=%.1f us median-of-long=%.1f us' % (sys.argv[2][-40:], k, len(v), v[-1], sorted(x for x in v if x > v[-1] / 2)[len([x for x in v if x > v[-1] / 2]) // 2]))
PY
done
