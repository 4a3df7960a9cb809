#!/bin/bash
# kernel-trace stats of the long-chain workloads (runs on the GPU box via gpurun, from the repo root)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the snapshot root)}"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_long2
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o w -- python3 tools/prof_workloads.py long > $O/log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o w -- python3 tools/prof_workloads.py long >> $O/log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/prof_workloads.py long >> $O/log 2>&1
python3 tools/summarize_prof.py $O | grep -E "^dispatches|^pmc" | cut -c1-250
