#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
V=rosdyn_amd/variants
{
timeout 300 $K tsqr3 3 rosdyn_amd/librdyn_hip.so $V/librdyn_prio1.so $V/librdyn_prio3.so
timeout 300 $K tsqr2 3 rosdyn_amd/librdyn_hip.so $V/librdyn_prio1.so $V/librdyn_prio3.so
} > gpurun_out/r3/run17_kbench.txt 2>&1
