#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r4_run.sh <tag> [tests] -- GPU suite (or the named test files) + perf sheet + bench lines into gpurun_out/<tag>_*
T=${1:-r4}; shift
mkdir -p gpurun_out
python -m pytest ${@:-tests} -x -q -m gpu > gpurun_out/${T}_gputests.log 2>&1
tail -5 gpurun_out/${T}_gputests.log
python bench.py --sharded-legs > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 3000 gpurun_out/${T}_bench.json; tail -5 gpurun_out/${T}_bench.err
python bench.py --single-process --gpus 1 > gpurun_out/${T}_bench_single.json 2> gpurun_out/${T}_bench_single.err
tail -c 1500 gpurun_out/${T}_bench_single.json; tail -5 gpurun_out/${T}_bench_single.err
