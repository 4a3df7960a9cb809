#!/bin/bash
# tools/pmc_gram.sh <kbench workload> <lib[@KEY=VALUE]> <tag>: matrix-pipe / issue counters of one kbench workload (separate --pmc passes)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the repo copy)}"
W=${1:?kbench workload}; L=${2:?library}; TAG=${3:?tag}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p $R/gpurun_out/pmc_$TAG
FAILED=0
i=0
for c in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o w -- $R/tools/_build/kbench $W 1 $R/$L > $R/gpurun_out/pmc_$TAG/log$i.txt 2>&1 || { echo "pass $i failed: see gpurun_out/pmc_$TAG/log$i.txt"; FAILED=1; }
done
cd $R
[ "$FAILED" = "0" ] || exit 1
python3 tools/pmc_table.py gpurun_out/pmc_$TAG duo
