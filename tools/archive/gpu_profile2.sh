#!/bin/bash
# kernel-trace stats of the secondary workloads (gram, mixed-chain, stacked layout, torque/inertia/kinematics)
export TMPDIR=/tmp
for W in gram multi stacked torque; do
  OUT=gpurun_out/prof_r1_$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 tools/prof_workload.py $W > $OUT/log.txt 2>&1
  python3 tools/summarize_prof.py $OUT | grep -E "^dispatches|k_gram|k_local|k_rowpair|k_base|k_regressor" | cut -c1-260 | tee $OUT/summary.txt
done
