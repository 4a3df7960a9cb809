#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras --no-config4 "$@" > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-config4 "$@" > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-config4 "$@" > /dev/null 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
python3 tools/summarize_prof.py $OUT ${PMC_KEY:-} | tee $OUT/summary.txt
