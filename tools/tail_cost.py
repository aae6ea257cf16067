#!/usr/bin/env python3
"""tools/tail_cost.py — cycles per outer iteration of waves that only drain (a 64x64x1 frame: 64 samples
per wave, nothing to refill): what a step of a bounce chain costs once a wave runs dry, by R1_COOP_LANES.
Diagnostic (STATS build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):  # tools only: another build of the library (e.g. lib/librays1_tuning.so), chosen explicitly
    binding.set_lib_path(os.environ["R1_LIB"])

w = h = 64
sc = r1.create_large_scene(w, h)
rend = r1.Renderer(0)
rend.set_scene(sc)
for v in (binding.VARIANT_BVH_STATS, binding.VARIANT_STATS):
    for it in range(2):
        img, rays, secs = rend.render(r1.make_params(w, h, 1, 10001, variant=v))
    st = rend.last_stats()
    info = rend.launch_info()
    print(f"variant {v} coop {os.environ.get('R1_COOP_LANES', 'default')}: rays {rays} blocks {info['blocks']} iterations {st['wave_iterations']} "
          f"alive/iter {st['alive_lanes'] / st['wave_iterations']:.1f} cycles/iter {st['cycles_wave'] / st['wave_iterations']:.0f} "
          f"(refill {st['cycles_refill'] / st['wave_iterations']:.0f} hit {st['cycles_candidates'] / st['wave_iterations']:.0f} shade {st['cycles_shade'] / st['wave_iterations']:.0f}) "
          f"device ms {rend.last_timing()[0]:.3f}")
