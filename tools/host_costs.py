#!/usr/bin/env python3
"""tools/host_costs.py — host-side cost of the drop-in's benchmark(): r1_set_scene and r1_render wall
time per call (what the reference's Timer span, rayweek1.cpp:848 -> :891, covers).  Diagnostic."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1

from rays1bench_amd import binding

w, h, spp = 1200, 800, 10
rend = r1.Renderer(0)
pinned = binding.HostFrame(w, h) if "pinned" in sys.argv[1:] else None  # r1_host_alloc'd pixel buffer instead of pageable numpy memory
for name, mk in (("small", r1.create_small_scene), ("medium", r1.create_medium_scene), ("large", r1.create_large_scene)):
    sc = mk(w, h)
    img = pinned.image if pinned else np.zeros((h, w, 3), np.uint8)
    p = r1.make_params(w, h, spp, 10001)
    rend.set_scene(sc)
    rend.render_into(p, img)
    ts, tr, dev = [], [], []
    for i in range(10):
        t0 = time.perf_counter()
        rend.set_scene(sc)
        t1 = time.perf_counter()
        rays, secs = rend.render_into(p, img)
        t2 = time.perf_counter()
        ts.append(t1 - t0), tr.append(t2 - t1), dev.append(secs)
    print(f"{name}: set_scene {np.median(ts)*1e3:.3f} ms  render {np.median(tr)*1e3:.3f} ms (device {np.median(dev)*1e3:.3f} ms)  "
          f"benchmark() span {np.median(np.add(ts, tr))*1e3:.3f} ms = {rays/np.median(np.add(ts, tr))/1e6:.0f} mrays/s")
