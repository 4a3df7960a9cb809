#!/bin/bash
# Round-3 profile set (runs on the GPU box via gpurun): see profiles/r3/.  Under rocprofv3 the program itself follows `--`.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r3
mkdir -p $O
# 1. the DEFAULT bench command: plain, then under --kernel-trace --stats
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final/trace -o bench -- python3 bench.py > $O/bench_default_under_rocprof.json 2> $O/final_trace.log
python3 tools/summarize_prof.py $O/final | grep -E "^==|^timed region|k_local_sweep|k_rowpair|k_image_sweep|k_gram|k_regressor|k_base|k_components|k_local_ik|k_cholqr|k_tsqr|^dispatches" | cut -c1-250 > $O/bench_default_summary.txt
cp $(find $O/final/trace -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv
# 2. HBM traffic of the headline kernel (stacked) and of the per-sample layout: separate --pmc passes
PMC_KEY=regressor_stacked_n6_P60_N1000000 bash tools/gpu_profile.sh r3_stacked --placements 0 > $O/prof_stacked.log 2>&1
PMC_KEY=regressor_per_sample_n6_P60_N1000000 bash tools/gpu_profile.sh r3_persample --placements 0 --y-layout per_sample > $O/prof_persample.log 2>&1
# 3. the secondary workloads: kernel stats + traffic (config 5, real chains) or matrix-pipe counters (Gram, robust factor)
for w in config5 real; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/trace -o w -- python3 tools/prof_r3_workloads.py $w > $O/$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$w/pmc_fetch -o w -- python3 tools/prof_r3_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$w/pmc_write -o w -- python3 tools/prof_r3_workloads.py $w >> $O/$w.log 2>&1
  python3 tools/summarize_prof.py $O/$w | grep -E "^dispatches|^pmc" | cut -c1-250 > $O/${w}_summary.txt
done
for w in gram cholqr; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/trace -o w -- python3 tools/prof_r3_workloads.py $w > $O/$w.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/$w/pmc1 -o w -- python3 tools/prof_r3_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/$w/pmc2 -o w -- python3 tools/prof_r3_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$w/pmc3 -o w -- python3 tools/prof_r3_workloads.py $w >> $O/$w.log 2>&1
  python3 tools/summarize_prof.py $O/$w | grep -E "^dispatches" | cut -c1-250 > $O/${w}_summary.txt
  for d in pmc1 pmc2 pmc3; do python3 tools/pmc_table.py $O/$w/$d k_regressor >> $O/${w}_summary.txt; done
done
python3 tools/pmc_config5.py $O/config5 > $O/config5_traffic.txt 2>&1
ls $O
