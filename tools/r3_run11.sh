#!/bin/bash
# store-schedule A/B of the stacked kernel over 10 output allocations, two fresh processes
mkdir -p gpurun_out/r3
K=tools/_build/kbench
V=rosdyn_amd/variants
for p in 1 2; do
KB_BUFFERS=10 timeout 600 $K stacked 1 rosdyn_amd/librdyn_hip.so $V/librdyn_rot.so $V/librdyn_wg256.so $V/librdyn_wg256rot.so $V/librdyn_plain.so > gpurun_out/r3/run11_sched_p$p.txt 2>&1
done
