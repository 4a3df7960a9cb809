#!/bin/bash
# tools/isa.sh <file.hip in rosdyn_amd/csrc> <mangled-name substring> [extra flags]: registers, scratch and the opcode mix of the
# kernel's largest loop (absolute paths: safe from any working directory)
F=$1; K=$2; shift 2
cd /root/repo/rosdyn_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -S --cuda-device-only -Wno-cuda-compat -ffp-contract=on "$@" $F -o /tmp/isa_$$.s 2>&1 | grep -E "error" -A4
awk "/amdhsa_kernel .*$K/,/end_amdhsa_kernel/" /tmp/isa_$$.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" | tr '\n' ' '; echo
python3 - /tmp/isa_$$.s "$K" <<'PY'
import re,collections,sys
txt=open(sys.argv[1]).read().split('\n'); K=sys.argv[2]
start=[i for i,l in enumerate(txt) if re.match(r'^_Z\S*:',l) and K in l][0]
end=[i for i in range(start,len(txt)) if txt[i].strip().startswith('.Lfunc_end')][0]
body=txt[start:end]
allops=[x.strip().split()[0] for x in body if x.strip() and x.strip()[0] not in '.;' and not x.strip().endswith(':')]
print('kernel total', len(allops))
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
best=None
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)',l)
    if m:
        t=m.group(1) or m.group(2)
        if t in labels and labels[t]<i:
            a=labels[t]
            ops=[x.strip().split()[0] for x in body[a:i+1] if x.strip() and x.strip()[0] not in '.;' and not x.strip().endswith(':')]
            if best is None or len(ops)>best[2]: best=(a,i,len(ops),ops)
if best:
    print('largest loop', best[2], collections.Counter(best[3]).most_common(28))
PY
rm -f /tmp/isa_$$.s
