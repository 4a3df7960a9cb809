#!/bin/bash
mkdir -p gpurun_out/r3
tools/_build/store_bw 6 10 > gpurun_out/r3/run22_store_only.txt 2>&1
KB_BUFFERS=10 tools/_build/kbench stacked 1 rosdyn_amd/librdyn_hip.so > gpurun_out/r3/run22_kernel_stacked.txt 2>&1
KB_BUFFERS=10 tools/_build/kbench persample 1 rosdyn_amd/librdyn_hip.so > gpurun_out/r3/run22_kernel_persample.txt 2>&1
