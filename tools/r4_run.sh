#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r4_run.sh <tag> -- full GPU suite + perf sheet into gpurun_out/<tag>_*.log
T=${1:-r4}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.log 2>&1
tail -5 gpurun_out/${T}_gputests.log
python tools/perf_sheet.py > gpurun_out/${T}_perf_sheet.txt 2>&1
tail -60 gpurun_out/${T}_perf_sheet.txt
