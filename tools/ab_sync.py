#!/usr/bin/env python3
"""tools/ab_sync.py — A/B of synchronous frame time between env settings (each setting in its own
process, alternated; median device ms over frames).  usage: ab_sync.py "A=1 B=2" "A=0" ... [--variant V] [--reps N]"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = [a for a in sys.argv[1:] if not a.startswith("--")]
variant = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--variant=")), 4)
reps = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--reps=")), 5)
child = f"""
import sys, statistics
sys.path.insert(0, {ROOT!r})
import os
import rays1bench_amd as r1
if os.environ.get("R1_LIB"):  # the -DR1_TUNING build reads the R1_* knobs; the shipped library reads none
    from rays1bench_amd import binding
    binding.set_lib_path(os.environ["R1_LIB"])
w, h, spp = 1200, 800, 10
rend = r1.Renderer(0); rend.set_scene(r1.create_large_scene(w, h))
p = r1.make_params(w, h, spp, 10001, variant={variant})
ms = []
for i in range(40):
    rend.render(p)
    if i >= 8: ms.append(rend.last_timing()[1])
print(statistics.median(ms), min(ms))
"""
res = {c: [] for c in cfgs}
for r in range(reps):
    for c in cfgs:
        env = dict(os.environ)
        for kv in c.split():
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        res[c].append(tuple(float(x) for x in out.stdout.split()))
for c in cfgs:
    med = [m for m, _ in res[c]]
    print(f"{c:50s} median-of-medians {statistics.median(med):.4f} ms   min {min(m for _, m in res[c]):.4f} ms   runs {' '.join('%.3f' % m for m in med)}")
