#!/bin/bash
mkdir -p gpurun_out/r3
K=tools/_build/kbench
L=rosdyn_amd/variants/librdyn_probes.so
{
for s in 2048 1024 512 256; do echo "== subsample tiles $s"; RDYN_CHOLQR_SUBTILES=$s timeout 300 $K tsqr3 2 $L@RDYN_TSQR_ROUTE=cholqr; RDYN_CHOLQR_SUBTILES=$s timeout 300 $K tsqr2 2 $L@RDYN_TSQR_ROUTE=cholqr; done
} > gpurun_out/r3/run16_kbench.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu -k "cholqr" 2>&1 | tail -3 > gpurun_out/r3/run16_tests.txt
( time python bench.py > gpurun_out/r3/run16_bench.json 2> gpurun_out/r3/run16_bench.err ) 2> gpurun_out/r3/run16_bench_time.txt
