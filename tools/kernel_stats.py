#!/usr/bin/env python3
"""tools/kernel_stats.py — runs the diagnostic (R1_VARIANT_STATS) build of the trace kernel and
prints where a wave spends its cycles and how full its lanes are.  Measurement tool."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):  # tools only: another build of the library (e.g. lib/librays1_tuning.so), chosen explicitly
    binding.set_lib_path(os.environ["R1_LIB"])

scene = sys.argv[1] if len(sys.argv) > 1 else "large"
w, h, spp = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (1200, 800, 10)))
rend = r1.Renderer(0)
sc = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}[scene](w, h)
rend.set_scene(sc)
for variant in (binding.VARIANT_PREFILTER, binding.VARIANT_STATS):
    img, rays, secs = rend.render(r1.make_params(w, h, spp, 10001, variant=variant))
    img, rays, secs = rend.render(r1.make_params(w, h, spp, 10001, variant=variant))
    print(f"variant {variant}: rays {rays}  device {secs*1e3:.3f} ms  {rays/secs/1e6:.1f} mrays/s  trace/total ms {rend.last_timing()}")
st = rend.last_stats()
info = rend.launch_info()
waves = info["blocks"] * 4
st["groups"] = info["groups"]
st["spheres_active"] = info["spheres_active"]
st["rays"] = rays
st["waves"] = waves
st["lane_utilisation_at_sweep"] = st["alive_lanes"] / (64.0 * st["wave_iterations"])
st["iterations_per_wave"] = st["wave_iterations"] / waves
st["ideal_iterations_per_wave"] = rays / 64.0 / waves
for k in ("cycles_refill", "cycles_pass1", "cycles_candidates", "cycles_shade"):
    st["share_" + k[7:]] = st[k] / st["cycles_wave"]
st["cycles_per_iteration"] = st["cycles_wave"] / st["wave_iterations"]
st["pass1_cycles_per_iteration"] = st["cycles_pass1"] / st["wave_iterations"]
st["mean_wave_cycles"] = st["cycles_wave"] / waves
st["wave_slot_utilisation"] = st["mean_wave_cycles"] / max(st["span_cycles"], 1)
st["candidates_per_ray"] = st["candidates"] / rays
st["candidate_trips_per_iteration"] = st["candidate_loop_trips"] / st["wave_iterations"]
print(json.dumps(st, indent=1))
