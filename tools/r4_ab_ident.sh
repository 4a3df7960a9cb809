#!/bin/bash
# interleaved A/B of library variants on the 7-joint identification R factor (tools/prof_ident.py timing mode)
for r in 1 2; do
  for LIB in rosdyn_amd/librdyn_hip.so rosdyn_amd/variants/librdyn_solo4.so rosdyn_amd/variants/librdyn_solo4ah3.so rosdyn_amd/variants/librdyn_solo3ah3.so rosdyn_amd/variants/librdyn_pairs2.so; do
    echo -n "$LIB  "; RDYN_LIB_PATH=$PWD/$LIB RDYN_PROF_TIME=1 python3 tools/prof_ident.py 2>/dev/null | grep "ms per call"
  done
done
