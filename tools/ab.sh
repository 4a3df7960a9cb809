#!/bin/bash
# interleaved A/B of library builds: tools/ab.sh <rounds> <bench args...> ; libs = rosdyn_amd/librdyn_hip*.so
R=$1; shift
for r in $(seq 1 $R); do
  for LIB in rosdyn_amd/librdyn_hip*.so; do
    RDYN_LIB_PATH=$PWD/$LIB python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LIB', round(d['roofline']['kernel_ms']*1e3,1), 'us', round(d['roofline']['achieved']), 'GB/s')"
  done
done
