#!/bin/bash
mkdir -p gpurun_out/r3
export RDYN_LIB_PATH=$PWD/rosdyn_amd/variants/librdyn_probes.so
RDYN_CHOLQR_ROUNDS=1 python tools/debug_cholqr2.py > gpurun_out/r3/run15_debug.txt 2>&1
RDYN_CHOLQR_ROUNDS=2 python tools/debug_cholqr2.py >> gpurun_out/r3/run15_debug.txt 2>&1
