#!/bin/bash
# interleaved A/B of library builds on the 7-joint identification R factor (panda link0 -> link7 + 7 friction components, N = 1e6):
# the stock library and every variant under rosdyn_amd/variants/ (tools/build_variant.sh <name> rdyn_cholqr.hip -D...), two rounds.
# profiles/r4/ident7_ab.txt: -DRDYN_CHOLQR_TWO_PAIRS, -DRDYN_CHOLQR_SOLO3, -DRDYN_CHOLQR_AHEAD=3
for r in 1 2; do
  for LIB in rosdyn_amd/librdyn_hip.so rosdyn_amd/variants/*.so; do
    [ -f $LIB ] || continue
    echo -n "$LIB  "; RDYN_LIB_PATH=$PWD/$LIB RDYN_PROF_TIME=1 python3 tools/prof_ident.py 2>/dev/null | grep "ms per call"
  done
done
