#!/bin/bash
# usage (GPU box, repo root): tools/r4_prof.sh <tag> <script> [args]: rocprofv3 kernel trace + stats of a python script, summary into gpurun_out/<tag>_kernel_stats.csv
T=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/prof_$T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -o $T -- python3 $R/"$@" > $R/gpurun_out/prof_$T/run.log 2>&1
F=$(find $R/gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/${T}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s  avg %10.1f us  total %6.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
