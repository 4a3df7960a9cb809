#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
V=rosdyn_amd/variants
KB_BUFFERS=10 timeout 600 $K stacked 1 rosdyn_amd/librdyn_hip.so $V/librdyn_sf2.so $V/librdyn_sf5.so > gpurun_out/r3/run13_sched.txt 2>&1
