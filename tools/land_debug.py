#!/usr/bin/env python3
"""tools/land_debug.py — a frame through the throughput entry point (tiles summed inside the trace kernel) against the synchronous one, tile by tile."""
import os, sys
import numpy as np
try:
    import torch  # before librays1: one HIP runtime per process
except Exception:
    torch = None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rays1bench_amd as r1
from rays1bench_amd import binding
if os.environ.get("R1_LIB"):
    binding.set_lib_path(os.environ["R1_LIB"])
w, h, spp = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (96, 64, 4)))
rend = r1.Renderer(0)
sc = r1.create_small_scene(w, h) if len(sys.argv) < 5 else r1.create_large_scene(w, h)
rend.set_scene(sc)
p = r1.make_params(w, h, spp, 10001)
ref = np.zeros((h, w, 3), np.uint8)
ref_rays, _ = rend.render_into(p, ref)
hf = binding.HostFrames(w, h, 1)
for rep in range(3):
    hf._all[:] = 7
    rend.render_async(p, hf)
    try:
        rend.sync()
    except Exception as e:
        print("sync:", e)
    img = hf.image(0)
    print(f"rep {rep}: rays {hf.rays(0)} (ref {ref_rays}); launch {rend.launch_info()}")
    tx, ty = (w + 31) // 32, (h + 31) // 32
    for t in range(tx * ty):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        a, b = img[y0:y0 + 32, x0:x0 + 32], ref[y0:y0 + 32, x0:x0 + 32]
        print(f"  tile {t}: {'ok' if (a == b).all() else 'DIFF'} untouched={int((a == 7).all())} nonzero_diff={int((a != b).sum())}")

# the device-resident form: dense tile block + count, then the strided assemble (what smoke() and the rank path do)
assert torch is not None
from rays1bench_amd import sharding
nbytes = binding.shard_block_bytes(p)
for rep in range(3):
    rec = torch.full((nbytes + sharding.RECORD_TRAILER,), 9, dtype=torch.uint8, device="cuda")
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rend.render_shard_device(p, rec.data_ptr(), rec.data_ptr() + nbytes, 0)
    rend.assemble_device_strided(p, rec.data_ptr(), nbytes + sharding.RECORD_TRAILER, out.data_ptr(), 0)
    try:
        rend.sync()
    except Exception as e:
        print("sync:", e)
    torch.cuda.synchronize()
    blk = rec.cpu().numpy()
    img = out.cpu().numpy()
    print(f"shard rep {rep}: rays {sharding.total_rays(rec, 1)} (ref {ref_rays}) nbytes {nbytes} launch {rend.launch_info()}")
    tx, ty = (w + 31) // 32, (h + 31) // 32
    for t in range(tx * ty):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        a, b = img[y0:y0 + 32, x0:x0 + 32], ref[y0:y0 + 32, x0:x0 + 32]
        tb = blk[t * 3072:(t + 1) * 3072].reshape(32, 32, 3)[:b.shape[0], :b.shape[1]]
        print(f"  tile {t}: image {'ok' if (a == b).all() else 'DIFF'}  block {'ok' if (tb == b).all() else 'DIFF'} block untouched={int((tb == 9).all())}")

# E1: the same entry point (dense tile block) writing into page-locked HOST memory
hf2 = binding.HostFrames(w, h, 1)
for rep in range(2):
    hf2._all[:] = 9
    rend.render_shard_device(p, hf2.ptr, hf2.ptr + hf2.rays_offset, 0)
    rend.sync()
    blk = hf2._all
    print(f"E1 rep {rep}: rays {hf2.rays(0)} (ref {ref_rays})")
    for t in range(tx * ty):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        b = ref[y0:y0 + 32, x0:x0 + 32]
        tb = blk[t * 3072:(t + 1) * 3072].reshape(32, 32, 3)[:b.shape[0], :b.shape[1]]
        print(f"  tile {t}: block {'ok' if (tb == b).all() else 'DIFF'} untouched={int((tb == 9).all())}")
# E2: device target again, but waiting before anything else is enqueued, and reading the block back with a plain copy
rec = torch.full((nbytes + sharding.RECORD_TRAILER,), 9, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
rend.render_shard_device(p, rec.data_ptr(), rec.data_ptr() + nbytes, 0)
rend.sync()
torch.cuda.synchronize()
blk = rec.cpu().numpy()
print(f"E2: rays {sharding.total_rays(rec, 1)} (ref {ref_rays})")
for t in range(tx * ty):
    x0, y0 = (t % tx) * 32, (t // tx) * 32
    b = ref[y0:y0 + 32, x0:x0 + 32]
    tb = blk[t * 3072:(t + 1) * 3072].reshape(32, 32, 3)[:b.shape[0], :b.shape[1]]
    print(f"  tile {t}: block {'ok' if (tb == b).all() else 'DIFF'} untouched={int((tb == 9).all())}")

# the counter allocation after a failing launch
import ctypes as C
hf2._all[:] = 9
rend.render_shard_device(p, hf2.ptr, hf2.ptr + hf2.rays_offset, 0)
rend.sync()
buf = np.zeros(16384, np.uint8)
have = C.c_size_t()
L = r1.lib()
L.r1_debug_dump_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
print("dump rc", L.r1_debug_dump_counters(rend._c, buf.ctypes.data, buf.nbytes, C.byref(have)), have.value)
wds = buf.view(np.uint32)
for name, off in (("set0", 1024), ("set1", 4096 + 1024)):
    print(name, "heads", [int(wds[(off + 128 * q) // 4]) for q in range(8)], "active", [int(wds[(off + 128 * (8 + q)) // 4]) for q in range(8)],
          "debug", [hex(int(wds[(off + 128 * (16 + q)) // 4])) for q in range(8)])
fr = 8192
print("frame_rays", int(buf[fr:fr + 8].view(np.uint64)[0]), "frame_left", int(wds[(fr + 8) // 4]))
print("countdowns", [int(wds[(fr + 128 + 128 * t) // 4]) for t in range(6)])
