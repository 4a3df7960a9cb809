#!/bin/bash
mkdir -p gpurun_out/r3
export RDYN_LIB_PATH=$PWD/rosdyn_amd/variants/librdyn_probes.so RDYN_TSQR_ROUTE=cholqr RDYN_CHOLQR_ROUNDS=1
python tools/debug_cholqr.py ur10_like.urdf base_link wrist_3_link 200000 > gpurun_out/r3/run7_debug.txt 2>&1
python tools/debug_cholqr.py panda_like.urdf link0 link7 200000 >> gpurun_out/r3/run7_debug.txt 2>&1
