#!/bin/bash
# round 3, run 1: image-kernel tests for the fixed-joint patterns + A/B of the tau pin / launch bounds + real-chain rates
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_image.py tests/test_gpu_multichain.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3/run1_tests.txt
K=tools/_build/kbench
L="rosdyn_amd/librdyn_hip.so rosdyn_amd/variants/librdyn_nopin.so rosdyn_amd/variants/librdyn_w2.so"
{
for w in stacked persample; do $K $w 3 $L; done
export KB_URDF=tests/fixtures/ur10_public.urdf KB_BASE=base_link
for t in wrist_3_link flange tool0; do for w in stacked persample; do echo "== ur10_public base_link -> $t"; KB_TOOL=$t $K $w 2 $L; done; done
export KB_URDF=tests/fixtures/ur10_like.urdf; echo "== ur10_like -> tool0"; for w in stacked persample; do KB_TOOL=tool0 $K $w 2 $L; done
export KB_URDF=tests/fixtures/panda_like.urdf KB_BASE=link0
for t in link7 link8 hand; do for w in stacked persample; do echo "== panda $t"; KB_TOOL=$t $K $w 2 rosdyn_amd/librdyn_hip.so; done; done
} > gpurun_out/r3/run1_kbench.txt 2>&1
