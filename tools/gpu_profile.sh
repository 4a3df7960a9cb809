#!/bin/bash
# tools/gpu_profile.sh -- the ONE profile script (runs on the GPU box via gpurun, from the repo root).  Under rocprofv3 the program itself
# follows `--` (never a shell or env hop); PMC counters are collected in their own passes (never together with traces).
#   tools/gpu_profile.sh bench <tag> [bench args]   kernel-trace stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py;
#                                                   PMC_KEY=<key> updates profiles/pmc_latest.json (read by bench.py for roofline.traffic)
#   tools/gpu_profile.sh round <rN>                 the profile set of a round into gpurun_out/prof_<rN>/ (copy what is to be judged into
#                                                   profiles/<rN>/): the DEFAULT bench command plain and under --kernel-trace --stats, the
#                                                   headline kernel's HBM traffic (stacked, per-sample), the secondary workloads of
#                                                   tools/prof_workloads.py with kernel stats, HBM traffic (streaming ones) or matrix-pipe
#                                                   counters (Gram, robust factors)
#   tools/gpu_profile.sh script <tag> <script.py> [args]   kernel-trace stats of any python script; the per-kernel table is printed and
#                                                   copied to gpurun_out/<tag>_kernel_stats.csv
# (the per-round scripts of rounds 1-3 this replaces are in tools/archive/)
set -u
MODE=${1:-round}; shift || true
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$MODE" = "bench" ]; then
  TAG=${1:-bench}; shift || true
  OUT=gpurun_out/prof_$TAG
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras --no-config4 "$@" > $OUT/bench_trace.json 2> $OUT/trace.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-config4 "$@" > /dev/null 2> $OUT/pmc_fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-config4 "$@" > /dev/null 2> $OUT/pmc_write.log
  python3 tools/summarize_prof.py $OUT ${PMC_KEY:-} | tee $OUT/summary.txt
  exit 0
fi
if [ "$MODE" = "script" ]; then
  TAG=$1; shift
  OUT=gpurun_out/prof_$TAG
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 "$@" > $OUT/run.log 2>&1
  F=$(find $OUT -name "*kernel_stats.csv" | head -1)
  cp "$F" gpurun_out/${TAG}_kernel_stats.csv
  python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-90s calls %5s  avg %10.1f us  total %6.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  exit 0
fi
R=${1:-r4}
O=gpurun_out/prof_$R
mkdir -p $O
KERNELS="^==|^timed region|k_local_sweep|k_rowpair|k_image_sweep|k_gram|k_regressor|k_base|k_components|k_local_ik|k_cholqr|k_tsqr|k_pgram|^dispatches"
# 1. the DEFAULT bench command: plain, then under --kernel-trace --stats
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final/trace -o bench -- python3 bench.py > $O/bench_default_under_rocprof.json 2> $O/final_trace.log
python3 tools/summarize_prof.py $O/final | grep -E "$KERNELS" | cut -c1-250 > $O/bench_default_summary.txt
cp $(find $O/final/trace -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv
# 2. HBM traffic of the headline kernel (stacked) and of the per-sample layout: separate --pmc passes
PMC_KEY=regressor_stacked_n6_P60_N1000000 bash tools/gpu_profile.sh bench ${R}_stacked --placements 0 > $O/prof_stacked.log 2>&1
PMC_KEY=regressor_per_sample_n6_P60_N1000000 bash tools/gpu_profile.sh bench ${R}_persample --placements 0 --y-layout per_sample > $O/prof_persample.log 2>&1
# 3. the secondary workloads: kernel stats + traffic (streaming kernels) or matrix-pipe counters (Gram, robust factors)
for w in config5 real long; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/trace -o w -- python3 tools/prof_workloads.py $w > $O/$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$w/pmc_fetch -o w -- python3 tools/prof_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$w/pmc_write -o w -- python3 tools/prof_workloads.py $w >> $O/$w.log 2>&1
  python3 tools/summarize_prof.py $O/$w | grep -E "^dispatches|^pmc" | cut -c1-250 > $O/${w}_summary.txt
done
for w in gram cholqr ident tsqr_rows; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/trace -o w -- python3 tools/prof_workloads.py $w > $O/$w.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/$w/pmc1 -o w -- python3 tools/prof_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/$w/pmc2 -o w -- python3 tools/prof_workloads.py $w >> $O/$w.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$w/pmc3 -o w -- python3 tools/prof_workloads.py $w >> $O/$w.log 2>&1
  python3 tools/summarize_prof.py $O/$w | grep -E "^dispatches" | cut -c1-250 > $O/${w}_summary.txt
  for d in pmc1 pmc2 pmc3; do python3 tools/pmc_table.py $O/$w/$d k_ >> $O/${w}_summary.txt; done
done
python3 tools/pmc_config5.py $O/config5 > $O/config5_traffic.txt 2>&1
ls $O
