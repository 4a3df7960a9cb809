#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_suite.sh <tag> [tests] -- GPU suite (or the named test files) + perf sheet into gpurun_out/<tag>_*
set -u -o pipefail
T=${1:-r5}; shift || true
mkdir -p gpurun_out
python -m pytest ${@:-tests} -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${T}_gputests.log
RC=${PIPESTATUS[0]}
tail -5 gpurun_out/${T}_gputests.log
if [ "$RC" != "0" ]; then echo "GPU suite failed (pytest exit $RC): no perf sheet"; exit $RC; fi
python tools/perf_sheet.py > gpurun_out/${T}_perf_sheet.txt 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${T}_perf_sheet.txt | tail -60
