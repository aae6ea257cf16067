#!/usr/bin/env python3
"""tools/submit_times.py — host time of every r1_render_async call over a few rounds of frames in flight (diagnostic:
where a short run's submission time goes).  usage: submit_times.py [SLOTS] [ROUNDS] [host|device]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
slots_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
to_host = (sys.argv[3] if len(sys.argv) > 3 else "host") == "host"
use_torch = "torch" in sys.argv[4:]   # torch streams instead of the contexts' own
use_ring = "ring" in sys.argv[4:]     # r1_timing_begin/_end around every round (fresh per-frame events), as bench.py's timed region
os.environ["GPU_MAX_HW_QUEUES"] = str(slots_n)
if use_torch or "importtorch" in sys.argv[4:]:  # importtorch: torch's bundled HIP runtime, but the contexts' own streams
    import torch
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):
    binding.set_lib_path(os.environ["R1_LIB"])

import ctypes as C


def raw_streams(n, prio=None):
    """n hipStream_t created directly on the runtime librays1 is linked to (dlsym through its handle)."""
    L = r1.lib()
    out = []
    for _ in range(n):
        st = C.c_void_p()
        if prio is None:
            rc = L.hipStreamCreateWithFlags(C.byref(st), C.c_uint(1))  # hipStreamNonBlocking
        else:
            rc = L.hipStreamCreateWithPriority(C.byref(st), C.c_uint(1), C.c_int(prio))
        assert rc == 0, rc
        out.append(st.value)
    return out


raw = None
if "rawpre" in sys.argv[4:]:
    raw = raw_streams(slots_n)
if "rawpre64" in sys.argv[4:]:
    raw = raw_streams(64)[:slots_n]
if "dummypre" in sys.argv[4:]:
    _dummy = raw_streams(slots_n)  # created, never used
w, h, spp = 1200, 800, 10
sc = r1.create_large_scene(w, h)
tile = [int(a[5:]) for a in sys.argv[4:] if a.startswith("tile=")]
tw, th = (tile[0], tile[0]) if tile else (32, 32)
p = r1.make_params(w, h, spp, 10001, tile_w=tw, tile_h=th)
ctx = []
extra = slots_n if "last" in sys.argv[4:] else 0  # "last": twice as many contexts, only the later half is used
if "split" in sys.argv[4:]:  # all contexts (and their streams) first, scenes afterwards
    rs = [r1.Renderer(0) for i in range(slots_n)]
    for c in rs:
        c.set_scene(sc)
        ctx.append((c, binding.HostFrame(w, h)))
else:
    for i in range(slots_n + extra):
        c = r1.Renderer(0)
        c.set_scene(sc)
        ctx.append((c, binding.HostFrame(w, h)))
    ctx = ctx[extra:]
if "dummypost" in sys.argv[4:]:
    _dummy = raw_streams(slots_n)  # created, never used
if "rawpost" in sys.argv[4:]:
    raw = raw_streams(slots_n)
if "rawpostprio" in sys.argv[4:]:
    raw = raw_streams(slots_n, 0)


class _S:
    def __init__(self, v):
        self.cuda_stream = v


streams = [torch.cuda.Stream() for _ in ctx] if use_torch else ([_S(v) for v in raw] if raw else [None] * len(ctx))
for rnd in range(rounds):
    ts = []
    if use_ring:
        for c, hf in ctx:
            c.timing_begin(3)
    t0 = time.perf_counter()
    for (c, hf), st in zip(ctx, streams):
        a = time.perf_counter()
        sp = st.cuda_stream if st is not None else None
        if to_host:
            c.render_async(p, hf, sp)
        else:
            c.render_frame_device(p, sp)
        ts.append((time.perf_counter() - a) * 1e3)
    sub = time.perf_counter() - t0
    if use_torch:
        torch.cuda.synchronize()
    elif raw or "devsync" in sys.argv[4:]:
        r1.lib().hipDeviceSynchronize()
    for c, hf in ctx:
        c.sync()
    tot = time.perf_counter() - t0
    if use_ring:
        for c, hf in ctx:
            c.timing_end()
    print(f"round {rnd}: submit {sub * 1e3:.3f} ms, all landed {tot * 1e3:.3f} ms ({tot / slots_n * 1e3:.4f} ms/frame); per call ms: "
          + " ".join(f"{t:.3f}" for t in ts))
