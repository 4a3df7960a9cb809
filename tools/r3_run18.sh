#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
V=rosdyn_amd/variants
{
timeout 300 $K stacked 2 $V/librdyn_nocopy5.so $V/librdyn_nocopy4.so $V/librdyn_nocopy1.so
KB_URDF=tests/fixtures/panda_like.urdf KB_BASE=link0 KB_TOOL=link7 timeout 300 $K stacked 2 $V/librdyn_nocopy7.so
} > gpurun_out/r3/run18_kbench.txt 2>&1
