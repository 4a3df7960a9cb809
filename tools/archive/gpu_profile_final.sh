#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of the DEFAULT bench command, plus the bench line itself.
# usage: tools/gpu_profile_final.sh <tag>
set -u
TAG=${1:-r1_final}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python3 tools/summarize_prof.py $OUT | grep -E "^==|^timed region|k_local_sweep|k_rowpair|k_image_sweep|k_gram|k_regressor|k_base|k_components|k_local_ik|^dispatches" | cut -c1-250 | tee $OUT/summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tail -1 $OUT/bench_plain.json
