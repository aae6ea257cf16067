#!/bin/bash
# tools/disasm.sh [LIB] — gfx950 disassembly of every code object of librays1.so (one per translation unit) into /tmp/r1_disasm.s
LIB=$(realpath ${1:-$(dirname $0)/../rays1bench_amd/lib/librays1.so})
python3 - "$LIB" <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(sys.argv[1])), "..", "..", "tools"))
sys.path.insert(0, "/root/repo/tools")
import kernel_meta
with tempfile.TemporaryDirectory() as tmp, open("/tmp/r1_disasm.s", "w") as out:
    for co in kernel_meta.code_objects(sys.argv[1], tmp):
        out.write(subprocess.run([kernel_meta.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout)
PY
wc -l /tmp/r1_disasm.s
