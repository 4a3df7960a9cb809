#!/bin/bash
# Round-2 profile set (runs on the GPU box via gpurun): kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the
# default (stacked) and per-sample bench commands, the full default command, and the PMC set of the wave-pair Gram kernel.
export TMPDIR=/tmp
PMC_KEY=regressor_stacked_n6_P60_N1000000 bash tools/gpu_profile.sh r2_stacked > gpurun_out/prof_r2_stacked.log 2>&1
PMC_KEY=regressor_per_sample_n6_P60_N1000000 bash tools/gpu_profile.sh r2_persample --y-layout per_sample > gpurun_out/prof_r2_persample.log 2>&1
bash tools/gpu_profile_final.sh r2_final > gpurun_out/prof_r2_final.log 2>&1
OUT=gpurun_out/prof_r2_duo
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 tools/prof_pipe.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc1 -o w -- python3 tools/prof_pipe.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $OUT/pmc2 -o w -- python3 tools/prof_pipe.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o w -- python3 tools/prof_pipe.py > $OUT/pmc3.log 2>&1
python3 tools/summarize_prof.py $OUT | grep -E "^dispatches|k_regressor|k_gram" | cut -c1-260 > $OUT/summary.txt
for d in pmc1 pmc2 pmc3; do python3 tools/pmc_table.py $OUT/$d k_regressor_gram >> $OUT/summary.txt; done
cat $OUT/summary.txt
tail -3 gpurun_out/prof_r2_stacked.log gpurun_out/prof_r2_persample.log
tail -c 1500 gpurun_out/prof_r2_final.log
