#!/bin/bash
mkdir -p gpurun_out/r3
( export RDYN_LIB_PATH=$PWD/rosdyn_amd/variants/librdyn_probes.so RDYN_TSQR_ROUTE=cholqr RDYN_CHOLQR_ROUNDS=1
python tools/debug_cholqr.py ur10_like.urdf base_link wrist_3_link 200000 > gpurun_out/r3/run8_debug.txt 2>&1
python tools/debug_cholqr.py panda_like.urdf link0 link7 1000000 >> gpurun_out/r3/run8_debug.txt 2>&1 )
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/run8_tests.txt
K=tools/_build/kbench
L=rosdyn_amd/variants/librdyn_probes.so
{
timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
timeout 300 $K tsqr3 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
KB_URDF=tests/fixtures/ur10_public.urdf KB_BASE=base_link KB_TOOL=tool0 timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
} > gpurun_out/r3/run8_kbench.txt 2>&1
