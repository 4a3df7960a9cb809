#!/bin/bash
# Round-2 secondary profile set: fp64-VALU share of the torque / inertia / kinematics kernels (what bounds them), refreshed Gram PMC
export TMPDIR=/tmp
OUT=gpurun_out/prof_r2_torque
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 tools/prof_workload.py torque > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o w -- python3 tools/prof_workload.py torque > $OUT/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o w -- python3 tools/prof_workload.py torque > $OUT/pmc2.log 2>&1
python3 tools/summarize_prof.py $OUT | grep -E "^dispatches" | cut -c1-260 > $OUT/summary.txt
for d in pmc1 pmc2; do python3 tools/pmc_table.py $OUT/$d "k_" >> $OUT/summary.txt; done
OUT=gpurun_out/prof_r2_duo2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 tools/prof_pipe.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o w -- python3 tools/prof_pipe.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o w -- python3 tools/prof_pipe.py > $OUT/pmc3.log 2>&1
python3 tools/summarize_prof.py $OUT | grep -E "^dispatches|k_regressor|k_gram" | cut -c1-260 > $OUT/summary.txt
for d in pmc1 pmc3; do python3 tools/pmc_table.py $OUT/$d k_regressor_gram >> $OUT/summary.txt; done
cat gpurun_out/prof_r2_torque/summary.txt gpurun_out/prof_r2_duo2/summary.txt
