#!/usr/bin/env python3
"""tools/tree_crossover.py — the exhaustive sweep against the box tree (= R1_VARIANT_DEFAULT, for every scene since round 4: this
table is the evidence; round 3 timed scenes of 9..127 spheres when they were set): slices of the large scene (the ground, the three big balls and the
first n lattice spheres) at 1200x800x10, one synchronous frame each through R1_VARIANT_PREFILTER and R1_VARIANT_BVH
(median device ms of 12 frames), plus image equality."""
import ctypes as C
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):  # tools only: another build of the library (e.g. lib/librays1_tuning.so), chosen explicitly
    binding.set_lib_path(os.environ["R1_LIB"])

w, h, spp = 1200, 800, 10
src = r1.create_large_scene(w, h)
arr = src.arrays()
cam = src.camera.contents
active = np.nonzero(arr["inv_radius"] != 0)[0]
big, lattice = active[-4:], active[:-4]
rend = r1.Renderer(0)


def both(label):
    res = {}
    for name, v in (("sweep", binding.VARIANT_PREFILTER), ("tree", binding.VARIANT_BVH)):
        p = r1.make_params(w, h, spp, 10001, variant=v)
        ms = []
        for i in range(14):
            out = np.zeros((h, w, 3), np.uint8)
            rays, _ = rend.render_into(p, out)
            if i >= 2:
                ms.append(rend.last_timing()[1])
        res[name] = (statistics.median(ms), rays, out)
    same = res["sweep"][1] == res["tree"][1] and (res["sweep"][2] == res["tree"][2]).all()
    rend.render_into(r1.make_params(w, h, spp, 10001), np.zeros((h, w, 3), np.uint8))
    default = {binding.VARIANT_PREFILTER: "sweep", binding.VARIANT_BVH: "tree"}[rend.launch_info()["kernel"]]
    faster = "tree" if res["tree"][0] < res["sweep"][0] else "sweep"
    behind = (res[default][0] / res[faster][0] - 1.0) * 100.0
    print(f"{label}: sweep {res['sweep'][0]:.3f} ms  tree {res['tree'][0]:.3f} ms  -> {faster}  DEFAULT runs {default}"
          f"{'' if default == faster else f' ({behind:.1f} % behind)'}  (rays {res['tree'][1]}, identical {same})")


for name, make in (("small scene", r1.create_small_scene), ("medium scene", r1.create_medium_scene)):
    rend.set_scene(make(w, h))
    both(name)
for n in (1, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 480):
    keep = np.concatenate([lattice[:: max(1, len(lattice) // n)][:n], big])
    sub = {k: np.ascontiguousarray(v[keep]) for k, v in arr.items()}
    pad = (-len(keep)) % 8
    for k in sub:
        fill = {"center_x": 999999999.0, "center_y": 999999999.0, "center_z": 999999999.0, "mat_type": 255}.get(k, 0)
        sub[k] = np.concatenate([sub[k], np.full(pad, fill, sub[k].dtype)])
    cs = binding.CScene()
    cs.count = len(sub["center_x"])
    for k, v in sub.items():
        if k != "mat_type":
            setattr(cs, k, v.ctypes.data_as(C.POINTER(C.c_float)))
    cs.mat_type = sub["mat_type"].ctypes.data_as(C.POINTER(C.c_uint8))
    rend.set_scene_raw(cs, cam)
    both(f"{len(keep):4d} spheres")
