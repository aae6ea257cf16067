#!/bin/bash
# tools/kernel_meta.sh — register / LDS / spill metadata of every kernel in librays1.so's gfx950 code object
set -e
LIB=$(realpath ${1:-$(dirname $0)/../rays1bench_amd/lib/librays1.so})
T=$(mktemp -d)
cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB >/dev/null 2>&1 || true
/opt/rocm/bin/roc-obj-ls $LIB 2>/dev/null | grep gfx950 | awk '{print $NF}' | head -1 > uri.txt || true
if [ -s uri.txt ]; then /opt/rocm/bin/roc-obj-extract "$(cat uri.txt)" >/dev/null 2>&1 || true; fi
CO=$(ls *.co 2>/dev/null | head -1)
if [ -z "$CO" ]; then
  # fall back: unbundle from the .hip_fatbin section
  objcopy -O binary --only-section=.hip_fatbin $LIB fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.co
  CO=k.co
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\n\s+- \.|\Z)', txt, re.S): pass
cur={}
for line in txt.splitlines():
    line=line.strip()
    for k in ('.name:','.vgpr_count:','.sgpr_count:','.vgpr_spill_count:','.sgpr_spill_count:','.group_segment_fixed_size:','.private_segment_fixed_size:','.agpr_count:'):
        if line.startswith(k) or line.startswith('- '+k):
            cur[k]=line.split(':',1)[1].strip()
    if line.startswith('.wavefront_size') or line.startswith('- .wavefront_size'):
        pass
    if ('.vgpr_spill_count:' in line):
        pass
import collections
# second pass: split on kernel blocks
blocks=re.split(r'\n\s*- \.agpr_count', txt)
for b in blocks[1:]:
    b='.agpr_count'+b
    g=lambda k:(re.search(re.escape(k)+r'\s*(\S+)', b) or [None,'?'])[1]
    print('%-60s vgpr %3s sgpr %3s vspill %3s sspill %3s lds %6s scratch %5s' % (g('.name:')[:60], g('.vgpr_count:'), g('.sgpr_count:'), g('.vgpr_spill_count:'), g('.sgpr_spill_count:'), g('.group_segment_fixed_size:'), g('.private_segment_fixed_size:')))
"
rm -rf $T
