#!/bin/bash
# tools/disasm.sh [LIB] — gfx950 disassembly of librays1.so's code object into /tmp/r1_disasm.s
LIB=$(realpath ${1:-$(dirname $0)/../rays1bench_amd/lib/librays1.so})
T=$(mktemp -d); cd $T
objcopy -O binary --only-section=.hip_fatbin $LIB fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn k.co > /tmp/r1_disasm.s
rm -rf $T
wc -l /tmp/r1_disasm.s
