#!/bin/bash
# round 3, run 4: preconditioned CholeskyQR route: tests + timing vs the Householder route
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/run4_tests.txt
K=tools/_build/kbench
L=rosdyn_amd/variants/librdyn_probes.so
{
timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
timeout 300 $K tsqr3 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
timeout 300 $K gram3 2 rosdyn_amd/librdyn_hip.so
KB_URDF=tests/fixtures/ur10_public.urdf KB_BASE=base_link KB_TOOL=tool0 timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr $L@RDYN_TSQR_ROUTE=householder
} > gpurun_out/r3/run4_kbench.txt 2>&1
