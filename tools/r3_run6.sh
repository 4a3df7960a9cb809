#!/bin/bash
mkdir -p gpurun_out/r3
python tools/debug_cholqr.py ur10_like.urdf base_link wrist_3_link 1000000 > gpurun_out/r3/run6_debug.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RDYN_TSQR_ROUTE=cholqr
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof6 -o t -- tools/_build/kbench tsqr3 1 rosdyn_amd/variants/librdyn_probes.so > gpurun_out/r3/run6_log.txt 2>&1
find gpurun_out/r3/prof6 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3/run6_kernel_stats.csv
