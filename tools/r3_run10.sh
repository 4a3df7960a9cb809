#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
{
timeout 300 $K tsqr3 3 rosdyn_amd/librdyn_hip.so rosdyn_amd/variants/librdyn_ah2.so rosdyn_amd/variants/librdyn_ah3.so
} > gpurun_out/r3/run10_kbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof10 -o t -- tools/_build/kbench tsqr3 1 rosdyn_amd/librdyn_hip.so > gpurun_out/r3/run10_log.txt 2>&1
find gpurun_out/r3/prof10 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3/run10_kernel_stats.csv
