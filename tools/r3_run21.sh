#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3/run21_tests.txt
K=tools/_build/kbench
V=rosdyn_amd/variants
{
timeout 300 $K tsqr3 3 rosdyn_amd/librdyn_hip.so $V/librdyn_before.so
timeout 300 $K tsqr2 3 rosdyn_amd/librdyn_hip.so $V/librdyn_before.so
KB_URDF=tests/fixtures/ur10_public.urdf KB_BASE=base_link KB_TOOL=tool0 timeout 300 $K tsqr2 2 rosdyn_amd/librdyn_hip.so $V/librdyn_before.so
} > gpurun_out/r3/run21_kbench.txt 2>&1
