#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_tsqr.py -x -q -m gpu -k "cholqr" 2>&1 | tail -12 > gpurun_out/r3/run14_tests.txt
