#!/usr/bin/env python3
"""tools/bvh_stats.py — traversal counters of the R1_VARIANT_BVH kernel (diagnostic build).
usage: bvh_stats.py SCENE W H SPP [GW GH]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):  # tools only: another build of the library (e.g. lib/librays1_tuning.so), chosen explicitly
    binding.set_lib_path(os.environ["R1_LIB"])

args = sys.argv[1:]
scene = args[0] if args else "large"
w, h, spp = (int(x) for x in (args[1:4] if len(args) > 3 else (1200, 800, 10)))
if scene == "grid":
    sc = r1.create_grid_scene(w, h, int(args[4]), int(args[5]))
else:
    sc = {"small": r1.create_small_scene, "medium": r1.create_medium_scene, "large": r1.create_large_scene}[scene](w, h)
rend = r1.Renderer(0)
rend.set_scene(sc)
for v in (binding.VARIANT_BVH, binding.VARIANT_BVH_STATS):
    for it in range(2):
        img, rays, secs = rend.render(r1.make_params(w, h, spp, 10001, variant=v))
    print(f"variant {v}: rays {rays} device {rend.last_timing()[1]:.3f} ms")
st = rend.last_stats()
leaf_lane_trips = st["leaf_lane_trips"]             # slot [14]: leaf trips summed over lanes; slot [5] ("cycles_pass1"): sphere-pair tests
info = rend.launch_info()
waves = info["blocks"] * 4
it = st["wave_iterations"]
out = {
    "rays": rays, "waves": waves, "wave_iterations": it,
    "lane_utilisation_at_sweep": st["alive_lanes"] / (64.0 * it),
    "root_steps_per_ray": st.get("root_steps", 0) / rays,  # the root's leaf + the box of its other child, outside the loops (counted as a leaf trip, not as a node visit)
    "node_visits_per_ray": st["candidates"] / rays,
    "sphere_tests_per_ray": st["cycles_pass1"] / rays,
    "node_loop_trips_per_iteration": st["candidate_loop_trips"] / it,
    "leaf_loop_trips_per_iteration": st["overflow_lanes"] / it,
    "node_loop_lane_utilisation": st["candidates"] / (64.0 * max(st["candidate_loop_trips"], 1)),
    "leaf_loop_lane_utilisation": leaf_lane_trips / (64.0 * max(st["overflow_lanes"], 1)),  # lanes on a leaf per leaf trip / 64
    "sphere_pairs_per_leaf_visit": st["cycles_pass1"] / max(leaf_lane_trips, 1),
    "share_refill": st["cycles_refill"] / st["cycles_wave"],
    "share_sweep": st["cycles_candidates"] / st["cycles_wave"],
    "share_shade": st["cycles_shade"] / st["cycles_wave"],
    "cycles_per_iteration": st["cycles_wave"] / it,
}
print(json.dumps(out, indent=1))
