#!/bin/bash
# round 3, run 2: reduced-chain Gram tests + Gram rates of the real chains
mkdir -p gpurun_out/r3
python -m pytest tests/test_reduction.py tests/test_gpu_gram.py tests/test_components.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3/run2_tests.txt
K=tools/_build/kbench
{
$K gram2 2 rosdyn_amd/librdyn_hip.so
export KB_URDF=tests/fixtures/ur10_public.urdf KB_BASE=base_link
for t in wrist_3_link tool0; do echo "== ur10_public base_link -> $t"; KB_TOOL=$t $K gram2 2 rosdyn_amd/librdyn_hip.so; KB_TOOL=$t $K ident 2 rosdyn_amd/librdyn_hip.so; done
export KB_URDF=tests/fixtures/ur10_like.urdf; echo "== ur10_like -> tool0"; KB_TOOL=tool0 $K gram2 2 rosdyn_amd/librdyn_hip.so
export KB_URDF=tests/fixtures/panda_like.urdf KB_BASE=link0
for t in link7 hand; do echo "== panda $t (N = 4e6)"; KB_TOOL=$t $K gram3 2 rosdyn_amd/librdyn_hip.so; done
} > gpurun_out/r3/run2_kbench.txt 2>&1
