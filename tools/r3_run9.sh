#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3/run9_tests.txt
K=tools/_build/kbench
L=rosdyn_amd/variants/librdyn_probes.so
{
timeout 300 $K tsqr3 3 $L@RDYN_TSQR_ROUTE=cholqr rosdyn_amd/variants/librdyn_wlds3.so
timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr
} > gpurun_out/r3/run9_kbench.txt 2>&1
