#!/usr/bin/env python3
"""tools/wave_timeline.py — when do the waves of ONE synchronous frame find the queue empty and when
do they end (STATS build of the tree kernel; 100 MHz device clock).  Diagnostic.
usage: wave_timeline.py [SCENE W H SPP]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):  # tools only: another build of the library (e.g. lib/librays1_tuning.so), chosen explicitly
    binding.set_lib_path(os.environ["R1_LIB"])

args = sys.argv[1:]
scene = args[0] if args else "large"
w, h, spp = (int(x) for x in (args[1:4] if len(args) > 3 else (1200, 800, 10)))
sc = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}[scene](w, h)
rend = r1.Renderer(0)
rend.set_scene(sc)
for it in range(3):
    img, rays, secs = rend.render(r1.make_params(w, h, spp, 10001, variant=binding.VARIANT_BVH_STATS))
log = rend.wave_log().astype(np.float64)
t0 = log[:, 0].min()
start, empty, end, iters = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0, (log[:, 2] - t0) / 100.0, log[:, 3]
print(f"{scene} {w}x{h}x{spp}: device {rend.last_timing()} ms, waves {len(log)}")
q = [0, 1, 10, 50, 90, 99, 100]
for name, v in (("start", start), ("queue found empty", empty), ("end", end), ("end - empty", end - empty), ("iterations", iters)):
    print(f"  {name:18s} us, percentiles {q}: " + " ".join(f"{np.percentile(v, p):9.1f}" for p in q))
# how many waves are alive over time
edges = np.linspace(0, end.max(), 41)
alive = [(int(((start <= t) & (end > t)).sum())) for t in edges]
print("  alive waves over time:", " ".join(str(a) for a in alive))
# per sub-queue (home = wave % 16) and per XCD (block % 8: blocks are dealt round-robin to the XCDs)
idx = np.arange(len(log))
for name, key in (("home sub-queue", ((idx // 32) * 4 + idx % 4) % 16), ("block % 8 (XCD)", (idx // 4) % 8)):
    print(f"  by {name}: median 'found empty' / median end / mean iterations")
    for g in np.unique(key):
        m = key == g
        print(f"    {g:3d}: {np.median(empty[m]):8.1f} {np.median(end[m]):8.1f} {iters[m].mean():7.1f}")
