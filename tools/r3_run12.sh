#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
V=rosdyn_amd/variants
KB_BUFFERS=10 timeout 600 $K stacked 1 rosdyn_amd/librdyn_hip.so $V/librdyn_wg256nb.so $V/librdyn_xcd.so $V/librdyn_wg256.so > gpurun_out/r3/run12_sched.txt 2>&1
