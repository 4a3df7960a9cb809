#!/bin/bash
# round 3, run 3: whole GPU suite + default bench line
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3/run3_tests.txt
python bench.py > gpurun_out/r3/run3_bench.json 2> gpurun_out/r3/run3_bench.err
tail -3 gpurun_out/r3/run3_bench.err
