#!/usr/bin/env python3
"""tools/burst_timeline.py DIR — start / end of the last N trace kernels (and anything else on the device) of a rocprofv3 --kernel-trace
run of `bench.py --steps 20 --warmup 5`: what a burst of twenty frames looks like on the device."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
trace = [r for r in rows if "r1_trace_kernel" in r["Kernel_Name"]]
last = trace[-n:]
t0 = int(last[0]["Start_Timestamp"])
t_end = max(int(r["End_Timestamp"]) for r in last)
print(f"{len(trace)} trace launches in the run; the last {n}: first start -> last end {(t_end - t0) / 1e6:.3f} ms")
for i, r in enumerate(last):
    print(f"  frame {i:2d}: start {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f}  end {(int(r['End_Timestamp']) - t0) / 1e6:7.3f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:7.3f} ms  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
others = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and "r1_trace_kernel" not in r["Kernel_Name"]]
print(f"other kernels after the burst's first start: {len(others)}")
for r in others[:12]:
    print(f"  {r['Kernel_Name'][:60]:60s} start {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
