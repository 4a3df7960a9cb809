#!/bin/bash
# usage: tools/kres_obj.sh <object or .so> [name filter]: registers / scratch / LDS of every gfx950 kernel inside a built object
# (reads the embedded code object: no recompilation)
F=$1; K=${2:-.}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$F" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    try: name=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name],capture_output=True,text=True).stdout.strip().split('(')[0]
    except Exception: pass
    agpr=blk.split()[0]
    print('%-70s vgpr %s agpr %s sgpr %s scratch %s lds %s'%(name[-70:],g('vgpr_count'),agpr,g('sgpr_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
" | grep -E "$K"
rm -rf $T
