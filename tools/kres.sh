#!/bin/bash
# usage: tools/kres.sh <file.hip> <kernel name substring> : compile to ISA, print resource usage + instruction mix
F=$1; K=$2
cd /root/repo/rosdyn_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -S --cuda-device-only $F -o /tmp/kres.s 2>&1 | grep -E "error" -A4
python3 /root/repo/tools/isa_mix.py /tmp/kres.s $K | head -1 | cut -c1-200
awk "/amdhsa_kernel .*$K/,/end_amdhsa_kernel/" /tmp/kres.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed|accum_offset" | tr '\n' ' '; echo
awk "/^_Z.*$K.*:/,/\.Lfunc_end/" /tmp/kres.s | grep -c "v_writelane\|v_readlane\|v_accvgpr" 
