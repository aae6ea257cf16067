#!/usr/bin/env python3
"""tools/kernel_meta.py [lib] [--json] — register / LDS / spill metadata and an instruction census (flat_load, scratch_, v_mfma)
of every kernel in every gfx950 code object of librays1.so.  The library is linked from several translation units, so its
.hip_fatbin section holds several offload bundles; each is unbundled and read with llvm-readelf / llvm-objdump.
Used by tests/test_host.py (no generic loads, no scratch, no MFMA in the product kernels) and by hand."""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(blob[s:e])
        co = os.path.join(tmp, f"k{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernels_of(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    meta = {}
    for b in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
        b = ".agpr_count" + b
        g = lambda k: (re.search(re.escape(k) + r"\s*(\S+)", b) or [None, "?"])[1]
        name = g(".name:")
        meta[name] = {"vgpr": g(".vgpr_count:"), "sgpr": g(".sgpr_count:"), "vgpr_spill": g(".vgpr_spill_count:"),
                      "sgpr_spill": g(".sgpr_spill_count:"), "lds": g(".group_segment_fixed_size:"),
                      "scratch": g(".private_segment_fixed_size:")}
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1) if m.group(1) in meta else None
            if cur:
                meta[cur].update({"flat_load": 0, "flat_store": 0, "scratch_insts": 0, "v_mfma": 0, "insts": 0, "lane_moves": 0})
            continue
        if cur:
            t = line.split()
            if len(t) < 1:
                continue
            op = t[0]
            meta[cur]["insts"] += 1
            if op.startswith("flat_load"):
                meta[cur]["flat_load"] += 1
            elif op.startswith("flat_store") or op.startswith("flat_atomic"):
                meta[cur]["flat_store"] += 1
            elif op.startswith("scratch_"):
                meta[cur]["scratch_insts"] += 1
            elif op.startswith("v_mfma"):
                meta[cur]["v_mfma"] += 1
            elif op in ("v_readlane_b32", "v_writelane_b32"):  # SGPR spill traffic: each one is a VALU issue slot
                meta[cur]["lane_moves"] += 1
    return meta


def collect(lib):
    with tempfile.TemporaryDirectory() as tmp:
        allk = {}
        for co in code_objects(lib, tmp):
            allk.update(kernels_of(co))
        return allk


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = os.path.realpath(args[0] if args else os.path.join(os.path.dirname(__file__), "..", "rays1bench_amd", "lib", "librays1.so"))
    k = collect(lib)
    if "--json" in sys.argv:
        print(json.dumps(k, indent=1, sort_keys=True))
        return
    for name in sorted(k):
        m = k[name]
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(R1TraceArgs)", "")[:58]
        print("%-58s vgpr %3s sgpr %3s vspill %3s sspill %3s lds %6s scratch %4s flat_ld %3s scratch_i %3s mfma %s lane_moves %s" % (
            short, m["vgpr"], m["sgpr"], m["vgpr_spill"], m["sgpr_spill"], m["lds"], m["scratch"], m.get("flat_load", "?"),
            m.get("scratch_insts", "?"), m.get("v_mfma", "?"), m.get("lane_moves", "?")))


if __name__ == "__main__":
    main()
