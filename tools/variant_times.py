#!/usr/bin/env python3
"""tools/variant_times.py — device time of one frame per kernel variant (measurement tool).
usage: variant_times.py SCENE W H SPP [GW GH] [--variants 0,4]   SCENE = small|medium|large|grid"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1

args = [a for a in sys.argv[1:] if not a.startswith("--")]
variants = [0, 4]
for a in sys.argv[1:]:
    if a.startswith("--variants"):
        variants = [int(x) for x in a.split("=")[1].split(",")]
scene = args[0] if args else "large"
w, h, spp = (int(x) for x in (args[1:4] if len(args) > 3 else (1200, 800, 10)))
if scene == "grid":
    sc = r1.create_grid_scene(w, h, int(args[4]), int(args[5]))
else:
    sc = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}[scene](w, h)
rend = r1.Renderer(0)
rend.set_scene(sc)
ref = None
for v in variants:
    best = 1e9
    for it in range(3):
        img, rays, secs = rend.render(r1.make_params(w, h, spp, 10001, variant=v))
        trace_ms, total_ms = rend.last_timing()
        best = min(best, total_ms)
    same = "" if ref is None else ("  image==first" if img.tobytes() == ref[0].tobytes() and rays == ref[1] else "  IMAGE DIFFERS")
    if ref is None:
        ref = (img, rays)
    print(f"{scene} {w}x{h}x{spp} variant {v}: rays {rays}  device {best:.3f} ms  {rays/best/1e3:.1f} mrays/s  info {rend.launch_info()}{same}", flush=True)
