#!/bin/bash
# VGPRs / SGPRs / scratch / LDS of the kernels of the built objects (code-object metadata notes):  bash tools/kernel_resources.sh [name filter] [object ...]
FLT=${1:-}; shift
OBJS=${@:-youreditableavatar_amd/lib/*.o}
for O in $OBJS; do
  T=$(mktemp -d)
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O 2>/dev/null
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle 2>/dev/null
  [ -s $T/dev.co ] && /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | FLT="$FLT" python3 -c "
import os, re, subprocess, sys
flt = os.environ.get('FLT', '')
for blk in sys.stdin.read().split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    dn = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dn: continue
    print(f\"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {dn[:110]}\")
"
  rm -rf $T
done
