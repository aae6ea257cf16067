#!/usr/bin/env python3
"""tools/mixed_variants.py — frames in flight through the tree and through the exhaustive sweep in ONE process and the same contexts:
does a phase leave something behind that slows the next?  (ms per frame, 20 contexts, 40 frames per phase.)"""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
USE_TORCH = "--torch" in sys.argv
if USE_TORCH:
    import torch  # before librays1: one HIP runtime per process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rays1bench_amd as r1
from rays1bench_amd import binding
w, h, spp = 1200, 800, 10
sc = r1.create_large_scene(w, h)
K = 20
rends = [r1.Renderer(0) for _ in range(K)]
for r in rends:
    r.set_scene(sc)
hfs = [binding.HostFrames(w, h, 1) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)] if USE_TORCH else [None] * K
sp = [s_.cuda_stream if s_ is not None else None for s_ in streams]


def sync_all():
    if USE_TORCH:
        torch.cuda.synchronize()
    else:
        for r in rends:
            r.sync()



def phase(name, variant, frames=40, host=True):
    p = r1.make_params(w, h, spp, 10001, variant=variant)
    for k in range(K):
        rends[k].render_async(p, hfs[k], sp[k]) if host else rends[k].render_frame_device(p, sp[k])
    sync_all()
    t0 = time.perf_counter()
    for f in range(frames):
        k = f % K
        rends[k].render_async(p, hfs[k], sp[k]) if host else rends[k].render_frame_device(p, sp[k])
    sync_all()
    dt = time.perf_counter() - t0
    print(f"{name:34s} {dt / frames * 1e3:.3f} ms per frame  ({hfs[0].rays(0)} rays)")


phase("tree", 0)
phase("sweep", binding.VARIANT_PREFILTER)
phase("sweep again", binding.VARIANT_PREFILTER)
phase("tree", 0)
phase("tree, frames left on the device", 0, host=False)
phase("sweep", binding.VARIANT_PREFILTER)
phase("sweep, frames left on the device", binding.VARIANT_PREFILTER, host=False)
