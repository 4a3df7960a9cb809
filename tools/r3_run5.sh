#!/bin/bash
mkdir -p gpurun_out/r3/prof5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RDYN_TSQR_ROUTE=cholqr
rocprofv3 --kernel-trace --stats -d gpurun_out/r3/prof5 -o tsqr3 -- tools/_build/kbench tsqr3 1 rosdyn_amd/variants/librdyn_probes.so > gpurun_out/r3/run5_log.txt 2>&1
find gpurun_out/r3/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3/run5_kernel_stats.csv
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/run5_tests.txt
