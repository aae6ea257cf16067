#!/usr/bin/env python3
"""tools/collect_profiles.py TAG — turns gpurun_out/prof_TAG (tools/profile_round.sh) into the JSON summaries of
profiles/TAG/ and profiles/pmc_traffic.json (what bench.py quotes, with its source)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", tag)


def means(c, match):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


tp, lat, pix, sw = ("r1_trace_kernel<4, false, false, 0>", "r1_trace_kernel<4, false, false, 1>", "r1_trace_kernel<4, false, false, 2>",
                    "r1_trace_kernel<2, false, false, 0>")
cmd = "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --inflight 1 (tools/profile_round.sh); mean per dispatch"
counters = {"command": cmd, "workload": "large 1200x800x10",
            "tree kernel, frames in flight (MODE 0)": {k: v[0] for k, v in {**means("pmc_sq1", tp), **means("pmc_sq2", tp)}.items()},
            "tree kernel, synchronous frame (MODE 1, 1536 workgroups)": {k: v[0] for k, v in {**means("pmc_sq1", lat), **means("pmc_sq2", lat)}.items()},
            "exhaustive sweep, frames in flight (MODE 0)": {k: v[0] for k, v in {**means("pmc_sq1", sw), **means("pmc_sq2", sw)}.items()},
            "note": "SQ_INSTS_VALU = VALU wave-instructions per launch (one frame); round 1's tree kernel: 888 M."}
json.dump(counters, open(f"{dst}/pmc_counters_tree_kernel.json", "w"), indent=1)


# ---- the same counters at the TIMED configuration (20 frames in flight) + how much the dispatches overlapped under collection
def overlap_of(c, match):
    """sum of the kernel's durations / span from its first start to its last end, from the kernel trace of the same run"""
    rows = []
    for f in glob.glob(f"{out}/{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not rows:
        return None
    rows.sort()
    rows = rows[len(rows) // 3:]  # skip the setup and warm-up launches
    busy = sum(b - a for a, b in rows)
    return {"dispatches": len(rows), "mean_duration_ms": busy / len(rows) / 1e6, "overlap": busy / max(rows[-1][1] - rows[0][0], 1)}


c20 = {**means("pmc20_sq1", tp), **means("pmc20_sq2", tp)}
if c20:
    v = {k: x[0] for k, x in c20.items()}
    ov = overlap_of("pmc20_sq2", tp) or overlap_of("pmc20_sq1", tp)
    derived = {}
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v:
        # SQ_* cycle counters count quad-cycles summed over waves (MI355X_MICROARCH.md: s_memtime tick vs SQ PMC units)
        derived["valu_share_of_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
        derived["wait_any_share_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"]
        derived["wait_inst_any_share_of_wave_cycles"] = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
    if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v:
        derived["valu_active_lane_fraction"] = v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v:
        # a SIMD issues one VALU instruction of one wave at a time: busy VALU quad-cycles summed over waves / (SQ busy cycles x 4 SIMDs per CU ... per SE)
        derived["valu_quadcycles_per_sq_busy_cycle"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_BUSY_CYCLES"]
    json.dump({"command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 20 "
                          "(20 frames in flight: the timed configuration); mean per dispatch of r1_trace_kernel<4,false,false,0>",
               "workload": "large 1200x800x10", "counters": v, "dispatch_overlap_under_collection": ov, "derived": derived,
               "note": "rocprofv3 reads the counters per dispatch; `dispatch_overlap_under_collection` says whether the 20 frames in flight still "
                       "overlapped while it did (1.0 = one dispatch at a time: then the wait / busy figures describe an isolated launch, "
                       "whatever --inflight says, and only the instruction counts transfer to the timed run)"},
              open(f"{dst}/pmc_counters_tree_kernel_inflight20.json", "w"), indent=1)
    print("inflight 20:", derived, ov)
f, w, fp, wp = means("pmc_FETCH_SIZE", tp), means("pmc_WRITE_SIZE", tp), means("pmcpix_FETCH_SIZE", pix), means("pmcpix_WRITE_SIZE", pix)
fl, wl, fs, ws = means("pmc_FETCH_SIZE", lat), means("pmc_WRITE_SIZE", lat), means("pmc_FETCH_SIZE", sw), means("pmc_WRITE_SIZE", sw)
hb = lambda a, b: (2 * a["FETCH_SIZE"][0] + b["WRITE_SIZE"][0]) * 1024
traffic = {"workload": "large 1200x800x10", "kernel": "r1_trace_kernel<4,false,false,0> (box tree, frames in flight: the kernel bench.py times)",
           "kernel_variant": 4,
           # the build the counters belong to: bench.py quotes them only while it runs the same library.  The
           # profile run writes the hash next to its outputs (tools/profile_round.sh: lib_sha16.txt)
           "lib_sha16": (open(f"{out}/lib_sha16.txt").read().strip() if os.path.exists(f"{out}/lib_sha16.txt") else None),
           "source": f"profiles/{tag}/pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --no-cpu-baseline "
                     f"--steps 3 --warmup 1 --inflight 1; mean over {f['FETCH_SIZE'][1]} dispatches)",
           "FETCH_SIZE_KB": f["FETCH_SIZE"][0], "WRITE_SIZE_KB": w["WRITE_SIZE"][0],
           "correction": "MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> x2; WRITE_SIZE taken as is",
           "hbm_bytes_per_launch": hb(f, w),
           "other_kernels": {"synchronous frame (MODE 1)": hb(fl, wl), "exhaustive sweep (MODE 0)": hb(fs, ws), "PIXEL mode (MODE 2, r1_set_pixel_mode)": hb(fp, wp)},
           "valu_wave_instructions_per_launch": counters["tree kernel, frames in flight (MODE 0)"]["SQ_INSTS_VALU"],
           "valu_source": f"profiles/{tag}/pmc_counters_tree_kernel.json (rocprofv3 --pmc SQ_INSTS_VALU ..., same command)",
           "valu_active_lane_fraction": counters["tree kernel, frames in flight (MODE 0)"]["SQ_THREAD_CYCLES_VALU"]
           / (counters["tree kernel, frames in flight (MODE 0)"]["SQ_ACTIVE_INST_VALU"] * 64),
           "note": "the trace kernel writes one 16-byte record per pixel-sample (9.6 M x 16 B = 153.6 MB) and reads the node table + sphere pairs through L1/L2: "
                   "HBM reads are noise.  The synchronous-frame kernel writes more (64-sample chunks per wave: more partial lines).  PIXEL mode writes the resolved pixels only."}
json.dump(traffic, open(f"{root}/profiles/pmc_traffic.json", "w"), indent=1)
json.dump(traffic, open(f"{dst}/pmc_hbm_traffic.json", "w"), indent=1)
ms = {str(k): json.loads(open(f"{out}/emulate_shards_{k}.json").read())["ms_per_step"] for k in (2, 4, 8)}
one = json.loads(open(f"{out}/bench_line.json").read().strip().splitlines()[-1])["ms_per_step"]
ms["1"] = one
short = {}
for k in (2, 4, 8):
    try:
        short[str(k)] = json.loads(open(f"{out}/emulate_shards_{k}_steps20.json").read())["ms_per_step"]
    except Exception:
        pass
try:
    short["1"] = json.loads(open(f"{out}/bench_line_driver_command.json").read().strip().splitlines()[-1])["ms_per_step"]
except Exception:
    pass
em = {"command": "python bench.py --steps 320 --warmup 32 --no-cpu-baseline --no-extras --emulate-shards N --rccl-selftest (one GPU carries rank 0's tiles of an N-GPU run through "
                 "the rank path: frame batches, one RCCL all-gather per launch through a one-rank communicator, copies to the host); N = 1: the bench line",
      "ms_per_frame_steps20": short,
      "ms_per_frame": ms, "fraction_of_ideal": {k: one / int(k) / v for k, v in ms.items() if k != "1"},
      "note": "single-GPU emulation of the per-rank load; the N-GPU run itself is the driver's"}
json.dump(em, open(f"{dst}/emulated_shards.json", "w"), indent=1)
print("hbm bytes per launch", traffic["hbm_bytes_per_launch"], "valu", traffic["valu_wave_instructions_per_launch"], traffic["other_kernels"])
print(em["ms_per_frame"], em["fraction_of_ideal"])
