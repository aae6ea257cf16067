#!/usr/bin/env python3
"""bench.py — mrays/s of the step13 hot path on N MI355X (BASELINE.json metric).

    python bench.py                              # N = 1
    python bench.py --gpus N                     # launches itself: N fresh rank processes (torch.distributed.run), RCCL
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # the driver's form: the same ranks
    python bench.py --gpus N --multi inproc      # ONE process, N GPUs: r1_multi_* (ncclCommInitAll), one frame at a time

A "step" is one full frame of the workload, from dispatch to PIXELS AND RAY COUNT ON THE HOST — the span the
reference's Timer covers (rayweek1.cpp:848 -> :891): every rank traces + resolves the tiles it owns (tile t -> rank
t % N, include/rays1.h r1_params.shard); at N > 1 the per-rank records (dense tile block + 8-byte ray count) are
all-gathered with RCCL (torch.distributed "nccl") — ONE collective per frame — and assembled, and the row-major image and the
frame's ray count are copied into page-locked host memory on the frame's own stream.  At N = 1 a frame is ONE launch: the trace
kernel sums every 32 x 32 tile on the XCD that traced it and stores the pixels and the ray count straight into the page-locked
frame (no resolve launch, no copy).  Several frames are in flight (one stream + context + host buffer each),
and the timed region ends when every frame has landed.
The frame is fixed (BASELINE: large scene, 1200x800x10 spp), so scaling is STRONG.  Inputs (sphere tables, camera)
are resident in HBM before the timed region.  `value_device_resident` is the same run with the frames left in HBM
(round 2's headline); `value_dispatch_to_host` is ONE synchronous frame at a time through r1_render() — what the
drop-in's benchmark() prints.  Data is synthetic by construction: the scenes are code, not files.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE_KIND = {"small": 0, "medium": 1, "large": 2, "grid": 3}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
VALU_ISSUE_PEAK_T = 1024 * 2.4e9 / 2 / 1e12  # wave-instructions/s: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles

# Frames in flight only overlap when their streams sit on different hardware queues; the ROCm default is 4.
# Measured on one MI355X: one queue per frame in flight is the optimum (16 in flight on 16 queues
# 1.18 ms per frame; 12 on 16: 1.77; 12 on 12: 1.20), 20 on 20 is 3 % better over a short run, and at 24 queues the
# process runs out of hardware queues (10-24 Grays/s).  Ranks that also run RCCL (its stream needs a queue of its own)
# keep 10 launches (of 2+ frames) in flight on 12 queues (run_ranks); one process driving N GPUs in-process runs one frame at a time.
INFLIGHT_SINGLE, QUEUES_SINGLE = 20, 20
INFLIGHT_RANK, QUEUES_RANK = 10, 12
INFLIGHT_RANK_BATCHED = 8   # launches in flight per rank when every launch carries a batch of N >= 3 frames
QUEUES_CLIFF = 24


def lib_sha16(path):
    """First 16 hex digits of the sha256 of a librays1 build (what profiles/pmc_traffic.json is keyed on)."""
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, w, h, spp):
    """Times the reference's own step13 scheduler + render_tile (oracle/_ref, kind
    "reference") on the host cores; falls back to the oracle port.  Bounded to ~10-20 s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import r1o

    def run(binary, runs):
        out = subprocess.run([binary, "bench", scene, str(w), str(h), str(spp), "0", str(runs)], capture_output=True, timeout=600)
        if out.returncode != 0:
            return None
        return [json.loads(l) for l in out.stdout.decode().strip().splitlines()]

    # the x86-64-v3 (AVX2 + FMA) build first: `-march=native` of the build container is another
    # machine's native on the GPU box
    for prefer_native in (False, True):
        binary = r1o.ref_binary(prefer_native)
        if not binary:
            continue
        try:
            probe = run(binary, 2)
        except Exception:
            probe = None
        if not probe:
            continue
        per = probe[-1]["seconds"]
        runs = max(3, min(40, int(12.0 / max(per, 1e-3))))
        rec = run(binary, runs + 1)
        if not rec:
            continue
        rec = rec[1:]  # first run pays page faults / thread start
        rays = sum(r["rays"] for r in rec)
        secs = sum(r["seconds"] for r in rec)
        threads = rec[0]["threads"]
        tiles = ((w + 31) // 32) * ((h + 31) // 32)
        return {"value": rays / secs / 1e6, "unit": "mrays/s", "cores": threads, "kind": "reference", "cpu_model": cpu_model(),
                "sample": f"{scene} {w}x{h}x{spp}: {len(rec)} frames through the reference's TileRenderScheduler/render_tile "
                          f"({os.path.basename(binary)}: flags of reference bench.py:175 with "
                          f"{'-march=x86-64-v3' if 'avx2' in os.path.basename(binary) else '-march=native of the build container'}), {secs:.1f} s",
                "note": f"the reference as it is: {threads} threads are spawned and joined for every {secs / len(rec) * 1e3:.0f} ms frame and share "
                        f"{tiles} tiles ({tiles / max(threads, 1):.1f} per thread), so at this frame size the number is mostly thread start-up "
                        "and imbalance, not the sweep (reported, not optimised against)"}
    # port: the oracle's restatement of the threaded path
    import rays1bench_amd as r1
    sc = r1.Scene(SCENE_KIND[scene], w, h)
    sa = r1o.SceneArrays.from_c(sc.spheres, sc.camera)
    t0, rays, n = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < 10.0 and n < 40:
        rays += r1o.render_threads(sa, w, h, spp, 0)[1]
        n += 1
    secs = time.perf_counter() - t0
    return {"value": rays / secs / 1e6, "unit": "mrays/s", "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{scene} {w}x{h}x{spp}: {n} frames through oracle/r1_oracle.c r1o_render_threads, {secs:.1f} s"}


def cpu_table(w, h, spp):
    """SURVEY.md §8f-4 within the rules (the CPU side lives in oracle/, reachable only from here):
    the reference's README-style table on this box's host cores — step13 multi-threaded and
    single-threaded (the reference's own TileRenderScheduler/render_tile, oracle/_ref) for the
    three scenes, and step1 (oracle port) on the small scene.  Bounded to a few seconds per entry."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import r1o
    rows = []
    binary = r1o.ref_binary(True) or r1o.ref_binary(False)
    for scene in ("small", "medium", "large"):
        for threads, (fw, fh) in ((0, (w, h)), (1, (max(w // 2, 1), max(h // 2, 1)))):
            if not binary:
                break
            try:
                out = subprocess.run([binary, "bench", scene, str(fw), str(fh), str(spp), str(threads), "3"], capture_output=True, timeout=300)
                rec = [json.loads(l) for l in out.stdout.decode().strip().splitlines()][1:]
                rows.append({"version": "step13", "scene": scene, "threads": rec[0]["threads"], "frame": f"{fw}x{fh}x{spp}",
                             "mrays_per_s": sum(r["rays"] for r in rec) / sum(r["seconds"] for r in rec) / 1e6, "kind": "reference"})
            except Exception as e:
                rows.append({"version": "step13", "scene": scene, "threads": threads, "error": str(e)})
    t0 = time.perf_counter()
    rays = r1o.step1_small(w // 2, h // 2, 1)[1]
    rows.append({"version": "step1", "scene": "small", "threads": 1, "frame": f"{w // 2}x{h // 2}x1",
                 "mrays_per_s": rays / (time.perf_counter() - t0) / 1e6, "kind": "port"})
    return rows


def cpu_baseline_grid(sc, w, h):
    """Scenes beyond the reference's MAX_SPHERES = 1024 (rayweek1.cpp:174) cannot go through its
    binaries: the oracle port's threaded exhaustive sweep, on a frame shrunk to ~10-30 s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import r1o
    sa = r1o.SceneArrays.from_c(sc.spheres, sc.camera)
    sw, sh, sspp = max(16, w // 16), max(9, h // 16), 1
    t0 = time.perf_counter()
    rays = r1o.render_threads(sa, sw, sh, sspp, 0)[1]
    secs = time.perf_counter() - t0
    if secs < 5.0:
        sspp = int(min(64, max(2, 15.0 / max(secs, 1e-3))))
        t0 = time.perf_counter()
        rays = r1o.render_threads(sa, sw, sh, sspp, 0)[1]
        secs = time.perf_counter() - t0
    return {"value": rays / secs / 1e6, "unit": "mrays/s", "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"same scene and camera, {sw}x{sh}x{sspp} frame through oracle/r1_oracle.c r1o_render_threads "
                      f"(exhaustive AVX2 sweep, the reference's algorithm), {secs:.1f} s"}


# flop per unit of the hit tests, counted from r1_trace.hpp (add / mul / compare / min / max = 1, fma = 2):
#   node visit   per child box (bvh_box): 3 fma, 3 fma, 3 sub + 2 max, 3 add + 2 min, 3 cmp = 25 -> 2 x 25 = 50 for the kernels that keep the
#                node table in LDS (the pad is evaluated once per ray since round 3); the big-scene kernels evaluate it
#                per node (fma + 3 mul): 55 (the round-1 form, which measured |m - o|^2 per box, was 2 x 32)
#   root step    the root's leaf (counted with the sphere-pair tests) + ONE box test, outside the walk's loops  = 25
#   exact_offer  pass 1 of Hitable::hit for one sphere = SURVEY.md §8d's 16 flop (3 sub, mul + 2 fma, mul + 2 fma, sub, mul + sub)
#   group test   7 fma + 1 sub of the prefilter (sweep_prefilter)                                                       = 15
FLOP_NODE, FLOP_NODE_BIG, FLOP_BOX, FLOP_SPHERE, FLOP_GROUP = 50.0, 55.0, 25.0, 16.0, 15.0


def measure_work(rend, p, info, binding):
    """Counts what one launch of the timed kernel executes: a synchronous frame through the
    diagnostic build of the same kernel (same samples; tools/bvh_stats.py, tools/kernel_stats.py)."""
    import numpy as np
    stats_variant = binding.VARIANT_BVH_STATS if info["kernel"] == 4 else binding.VARIANT_STATS
    q = binding.Params.from_buffer_copy(p)
    q.variant = stats_variant
    host = np.zeros((p.height, p.width, 3), np.uint8)
    rays, _ = rend.render_into(q, host)
    st = rend.last_stats()
    it = max(st["wave_iterations"], 1)
    if info["kernel"] == 4:
        # 64-bit counters each: [9] node visits, [5] ("cycles_pass1" of the sweep build) sphere-pair tests, [14] leaf trips x lanes
        visits, pairs, leaf_lane_trips = st["candidates"], st["cycles_pass1"], st["leaf_lane_trips"]
        roots = st.get("root_steps", 0)  # the root of the reference's scenes is visited outside the walk's loops: ONE box test + its leaf's pairs
        flop_node = FLOP_NODE_BIG if (info["spheres_active"] > 1023 or info["bvh_nodes"] > 256) else FLOP_NODE  # the big-scene kernels (r1_capi.cpp enqueue_frame)
        return {"source": "R1_VARIANT_BVH_STATS frame (same samples as the timed kernel)", "rays_per_launch": rays,
                "node_visits_per_ray": visits / rays, "sphere_pair_tests_per_ray": pairs / rays,
                "root_steps_per_ray": roots / rays, "flop_per_root_step": FLOP_BOX,
                "flop_per_node_visit": flop_node, "flop_per_sphere_pair_test": 2 * FLOP_SPHERE,
                "flop_per_launch": visits * flop_node + roots * FLOP_BOX + pairs * 2 * FLOP_SPHERE,
                "lane_utilisation": {"at_hit_test": st["alive_lanes"] / (64.0 * it),
                                     "node_loop": visits / (64.0 * max(st["candidate_loop_trips"], 1)),
                                     "leaf_loop": leaf_lane_trips / (64.0 * max(st["overflow_lanes"], 1))}}
    groups = info["groups"]
    slots = st["candidate_loop_trips"] * 64  # member slots the cooperative exact phase walked
    return {"source": "R1_VARIANT_STATS frame (same samples as the timed kernel)", "rays_per_launch": rays,
            "group_tests_per_ray": float(groups), "exact_slots_per_ray": slots / rays,
            "flop_per_group_test": FLOP_GROUP, "flop_per_exact_slot": FLOP_SPHERE,
            "flop_per_launch": rays * groups * FLOP_GROUP + slots * FLOP_SPHERE,
            "lane_utilisation": {"at_hit_test": st["alive_lanes"] / (64.0 * it)}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="large", choices=list(SCENE_KIND))
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=10)
    ap.add_argument("--grid", default="400x250", help="with --scene grid: small-sphere lattice WxH (BASELINE config 5: 400x250)")
    ap.add_argument("--seed", type=int, default=10001)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--multi", default="procs", choices=["procs", "inproc"],
                    help="N > 1: `procs` = one rank process per GPU, torch.distributed + RCCL (bench.py starts the ranks itself when it "
                         "was not started by torch.distributed.run); `inproc` = this one process drives all N GPUs through "
                         "r1_multi_* (ncclCommInitAll + one ncclAllGather per frame), one synchronous frame at a time")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight, each on its own stream + context workspace: later frames' workgroups fill "
                         "the CUs that a frame's last long bounce chains leave idle (1 = one frame at a time; "
                         f"0 = default: {INFLIGHT_SINGLE} on one GPU, {INFLIGHT_RANK} per rank when RCCL needs hardware queues of its own)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per launch (r1_render_batch_async: the persistent waves flow from one frame into the next, so a launch's "
                         "ramp and drain are paid once per batch); 0 = the mode's default")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this run (0 = the mode's measured default)")
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="tuning aid: render only shard 0 of K on one GPU, no collective (per-rank load of a K-GPU run)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="with --emulate-shards K on one GPU: also run the rank path's RCCL all-gather per frame through a ONE-rank "
                         "communicator (the collective's launch cost next to a 1/K frame; the data does not leave the GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--check", action="store_true", default=True,
                    help="(default) after the timed region compare the image and ray count that landed on the host with ONE synchronous, "
                         "unsharded r1_render of the same frame — another build of the kernel (latency mode), so `check: true` in the line "
                         "says the timed frames' pixels and count are the frame's")
    ap.add_argument("--no-check", dest="check", action="store_false")
    ap.add_argument("--pixel-mode", action="store_true",
                    help="r1_set_pixel_mode: lanes own pixels, 3 B/pixel written, no per-sample workspace, no resolve launch (~10 %% slower)")
    ap.add_argument("--no-host-copy", action="store_true", help="`value` = frames left in HBM (round 2's headline mode), no copies to the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (device-resident run, exhaustive sweep, r1_render span)")
    ap.add_argument("--lib", default="", help="another build of librays1.so (e.g. rays1bench_amd/lib/librays1_tuning.so, `make tuning`: "
                                              "the build that reads the R1_* tuning knobs); default: the shipped library")
    ap.add_argument("--cpu-table", action="store_true",
                    help="also time the reference's step13 (all threads / 1 thread) on the three scenes and the step1 port "
                         "on the host cores (cpu_baseline.table; adds ~30 s)")
    args = ap.parse_args(argv)
    if args.lib:
        args.lib = os.path.abspath(args.lib)  # (resolved where the command was given)
    return args


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH processes with torch.distributed.run
    (the form the driver uses) and pass their output and exit code through.  This parent never imports torch and never
    touches HIP — nothing that has initialised the GPU is ever exec'd or forked."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, cwd=os.getcwd())
    return proc.returncode


def set_hw_queues(args, default):
    q = args.hw_queues if args.hw_queues > 0 else default
    q = max(1, min(q, QUEUES_CLIFF - 1))
    os.environ["GPU_MAX_HW_QUEUES"] = str(q)  # explicit per mode; must be set before HIP starts
    return q


def workload_config(args, info, n, rays_per_step):
    w, h, spp = args.width, args.height, args.spp
    return {"workload": f"{args.scene} scene ({info['spheres_active']} spheres, N_pad {info['spheres_padded']}), {w}x{h}, {spp} spp, "
                        f"max 50 bounces, seed {args.seed}",
            "rays_per_step": rays_per_step, "tiles": "32x32, tile t -> rank t % N"}


def run_inproc(args):
    """ONE process, N GPUs: r1_multi_* — tile split, ncclCommInitAll, one ncclAllGather per frame, device 0 assembles and
    copies to the host (csrc/r1_multi.cpp; rayweek1.cpp:869-877: one call renders on all workers).  Frames in flight: K
    r1_multi objects (each with its own communicator and streams), r1_multi_render_async; `--inflight 1` = the synchronous
    r1_multi_render, one frame at a time (what rayweek1_hip --gather rccl does inside benchmark())."""
    n = args.gpus
    # r1_multi objects in flight.  One GPU: 8 (8 / 12 / 16 / 20: 32.3 / 31.8 / 31.9 / 31.9 Grays/s, profiles/r03/inproc_lanes.txt).  N > 1: TWO, each carrying
    # batches of N frames — every object owns N communicators (ncclCommInitAll) with their proxy threads and buffers, and collectives of different
    # communicators share the devices with persistent trace workgroups: 2 N communicators, not 8 N (N > 1 has not run on hardware)
    lanes = args.inflight if args.inflight > 0 else (8 if n == 1 else 2)
    set_hw_queues(args, lanes + 1 if lanes > 1 else 4)  # one hardware queue per frame in flight + one (torch's null stream); lanes + 2 is WORSE (mapping)
    import numpy as np
    import torch  # only for the contract's synchronize; r1_multi brings its own RCCL binding
    import rays1bench_amd as r1
    from rays1bench_amd import binding
    if args.lib:
        binding.set_lib_path(args.lib)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    if torch.cuda.device_count() < n:
        raise SystemExit(f"--multi inproc --gpus {n}: only {torch.cuda.device_count()} devices visible")
    w, h, spp = args.width, args.height, args.spp
    gw, gh = (int(v) for v in args.grid.split("x")) if args.scene == "grid" else (0, 0)
    scene = r1.Scene(SCENE_KIND[args.scene], w, h, gw, gh)
    # frames per launch: as for ranks, a device's share of one frame is a small launch at N >= 3 (run_ranks)
    B = args.batch if args.batch > 0 else (1 if n == 1 or lanes == 1 else max(1, min(n, args.steps // (2 * lanes))))
    multis = [binding.MultiRenderer(list(range(n))) for _ in range(lanes)]
    for m_ in multis:
        m_.set_scene(scene)
    hosts = [binding.HostFrames(w, h, B) for _ in range(lanes)]
    pending = [0] * lanes
    p = r1.make_params(w, h, spp, args.seed, variant=args.variant)
    img = np.zeros((h, w, 3), np.uint8)
    counter = [0]
    dev_s = [0.0]

    def step():
        k = (counter[0] // B) % lanes
        counter[0] += 1
        if lanes == 1:
            rays, s = multis[0].render_into(p, img)
            dev_s[0] += s
            return rays
        pending[k] += 1
        if pending[k] == B:
            multis[k].render_batch_async(p, B, hosts[k], 0)
            pending[k] = 0
        return None

    def fence():
        for k in range(lanes):  # a partial last batch
            if pending[k]:
                multis[k].render_batch_async(p, pending[k], hosts[k], 0)
                pending[k] = 0
        counter[0] = 0
        for m_ in multis:
            m_.sync()
        for d in range(n):
            torch.cuda.synchronize(d)

    for _ in range(2):  # setup: buffers, communicators, hardware queues
        for _ in range(lanes * B):
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    dev_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    submit = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if lanes > 1:
        rays_per_step = hosts[0].rays(0)
        img = hosts[0].image(0).copy()
    else:
        rays_per_step, _ = multis[0].render_into(p, img)
    info = multis[0].info()
    check = None
    if args.check:
        rend = r1.Renderer(0)
        rend.set_scene(scene)
        ref = np.zeros((h, w, 3), np.uint8)
        ref_rays, _ = rend.render_into(r1.make_params(w, h, spp, args.seed, variant=args.variant), ref)
        check = bool(ref.tobytes() == img.tobytes() and ref_rays == rays_per_step
                     and all(hf.rays(0) == ref_rays and hf.image(0).tobytes() == ref.tobytes() for hf in hosts[:min(lanes, args.steps)] if lanes > 1))
        rend.close()
    cfg = workload_config(args, info["first_device"], n, rays_per_step)
    cfg.update({"parallelism": f"tile-split x{n}, ONE process (r1_multi: ncclCommInitAll, one ncclAllGather per frame, RCCL {info['rccl_version']})",
                "value_mode": (f"{lanes * B} frames in flight = {lanes} r1_multi objects (own communicator + streams each) x {B} frames per launch, r1_multi_render[_batch]_async: every frame "
                               "ends with its pixels + ray count copied to page-locked HOST memory by device 0 (rayweek1.cpp:848 -> :891, pipelined)")
                              if lanes > 1 else
                              "one synchronous frame at a time through r1_multi_render: dispatch -> pixels + ray count on the host "
                              "(rayweek1.cpp:848 -> :891); latency-mode kernels, no frames in flight",
                "frames_in_flight": lanes * B, "launches_in_flight": lanes, "frames_per_launch": B, "communicators": lanes * n,
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "workgroups": info["first_device"]["blocks"], "threads_per_workgroup": info["first_device"]["threads_per_block"],
                "host_submit_ms_per_step": submit / args.steps * 1e3,
                "device_ms_per_step": dev_s[0] / args.steps * 1e3 if lanes == 1 else None})
    out = {"metric": f"mrays/s on '{args.scene}' scene {w}x{h}x{spp}spp", "value": rays_per_step * args.steps / elapsed / 1e6, "unit": "mrays/s",
           "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg,
           "roofline": {"bound": "valu", "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None,
                        "note": "the in-process mode reports the rate of the N-GPU frames; the kernel's roofline is measured by the default mode"}}
    if check is not None:
        out["check"] = check
    print(json.dumps(out), flush=True)
    for m_ in multis:
        m_.close()
    for hf in hosts:
        hf.close()
    return 0


def run_ranks(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = world
    uses_rccl = (n > 1 and args.backend == "nccl") or args.rccl_selftest
    shards_ = max(n, args.emulate_shards, 1)
    # Frames per launch (r1_render_shard_device_batch) and launches in flight for ranks.  Measured on one MI355X carrying
    # rank 0's tiles with the rank path's all-gather through a one-rank RCCL communicator (profiles/r03/batch_sweep.txt):
    #   * a rank's share of one frame is too small a launch at N >= 4 (8 shards: 23.3 Grays/s per rank with one frame per
    #     launch, 29.4 with 8; 4 shards: 28.2 -> 30.5; 2 shards: 32.0 / 31.1; whole frames at N = 1: 32.5 alone, 30-31 batched);
    #   * collectives issued from more than ~12 streams stall the submitting thread for ~7 ms every dozen launches (torch's
    #     process group + the runtime's lazily mapped hardware queues): a 20-step run on 16 streams spends 9 of its 11 ms there.
    # So ranks keep at most 10 launches in flight, of at least 2 frames each, and batch N frames per launch when the run is
    # long enough to keep ~16 launches busy (a 20-step run is better off with 10 small launches than with 3 large ones).
    if args.batch <= 0:
        args.batch = 1 if (shards_ == 1 or args.pixel_mode) else max(2, min(shards_, args.steps // 16))
    if args.inflight <= 0:
        args.inflight = INFLIGHT_SINGLE if shards_ == 1 else (INFLIGHT_RANK if args.batch <= 2 else INFLIGHT_RANK_BATCHED)
    extra = QUEUES_RANK - INFLIGHT_RANK if uses_rccl else 0
    args.inflight = max(1, min(args.inflight, QUEUES_CLIFF - 1 - extra))  # launches' queues + RCCL's stay under the cliff
    set_hw_queues(args, args.inflight + extra)
    import numpy as np
    import torch
    import rays1bench_amd as r1
    from rays1bench_amd import binding, sharding
    if args.lib:
        binding.set_lib_path(args.lib)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    if os.environ.get("R1_BENCH_DEVICE"):  # rehearsal: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["R1_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if n > 1 or args.rccl_selftest:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if n == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend != "nccl":
            dist.init_process_group(args.backend, rank=rank, world_size=n)
        else:
            try:
                dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", local_rank))
            except TypeError:  # older torch: no device_id keyword
                dist.init_process_group("nccl", rank=rank, world_size=n)

    w, h, spp = args.width, args.height, args.spp
    dev = torch.device("cuda", local_rank)
    shards = max(n, args.emulate_shards, 1)
    sharded = shards > 1
    p = r1.make_params(w, h, spp, args.seed, shard=rank if n > 1 else 0, num_shards=shards, variant=args.variant)
    block_bytes = binding.shard_block_bytes(p)
    record_bytes = binding.shard_record_bytes(p)  # tile block (padded to 8 bytes) + uint64 ray count: one all-gather per frame
    trailer = record_bytes - sharding.RECORD_TRAILER
    img_bytes = w * h * 3
    img_pad = (img_bytes + 7) & ~7
    host_copy = [not args.no_host_copy]

    B = max(1, args.batch)  # frames per launch (r1_render_batch_async): the waves flow from one frame into the next
    if args.pixel_mode:
        B = 1           # PIXEL mode has no batch form
    frec = binding.frame_record_bytes(p)  # image (padded to 8 bytes) + uint64 ray count

    submit_max = {"render": 0.0, "gather": 0.0, "assemble": 0.0, "copy": 0.0}

    class Slot:
        """One launch in flight: its own context (stream-ordered workspace), stream, device buffers and page-locked host
        frames.  step() adds a frame to the slot's pending batch; the batch is launched when it is full (or at a fence)."""

        def __init__(self):
            self.rend = r1.Renderer(local_rank)
            gw, gh = (int(v) for v in args.grid.split("x")) if args.scene == "grid" else (0, 0)
            self.scene = r1.Scene(SCENE_KIND[args.scene], w, h, gw, gh)
            self.rend.set_scene(self.scene)
            if args.pixel_mode:
                self.rend.set_pixel_mode(True)
            self.host = binding.HostFrames(w, h, B)  # pixels + ray counts land here
            self.host_t = torch.from_numpy(self.host._all)  # torch view of the page-locked frames: target of the non-blocking copies
            self.stream = torch.cuda.Stream(device=dev)
            self.pending, self.last = 0, 0
            if sharded or args.pixel_mode:
                self.records = torch.zeros(B * record_bytes, dtype=torch.uint8, device=dev)
                self.gathered = torch.zeros(n * B * record_bytes, dtype=torch.uint8, device=dev) if n > 1 else self.records
                self.frames = torch.zeros(B * frec, dtype=torch.uint8, device=dev)  # assembled frame records: one copy brings them to the host
                if args.rccl_selftest and n == 1:
                    self.selftest = torch.zeros(B * record_bytes, dtype=torch.uint8, device=dev)

        def step(self):
            self.pending += 1
            if self.pending == B:
                self.launch()

        def launch(self):
            k, self.pending = self.pending, 0
            if k == 0:
                return
            self.last = k
            sp = self.stream.cuda_stream
            if not (sharded or args.pixel_mode):
                # whole frames on this GPU: ONE launch that traces, sums its tiles and stores them into the page-locked frame records
                if B == 1:
                    self.rend.render_async(p, self.host, sp) if host_copy[0] else self.rend.render_frame_device(p, sp)
                else:
                    self.rend.render_batch_async(p, k, self.host if host_copy[0] else None, 0, sp)
                return
            with torch.cuda.stream(self.stream):
                t_ = [time.perf_counter()]
                if args.pixel_mode:
                    self.rend.render_shard_device(p, self.records.data_ptr(), self.records.data_ptr() + trailer, sp)
                else:
                    self.rend.render_shard_device_batch(p, k, self.records.data_ptr(), 0, sp)
                t_.append(time.perf_counter())
                if n > 1:  # the one exchange step of the batch: [rank][frame][record]
                    sharding.gather_records(dist, self.records[:k * record_bytes], self.gathered[:n * k * record_bytes])
                elif args.rccl_selftest:
                    dist.all_gather_into_tensor(self.selftest[:k * record_bytes], self.records[:k * record_bytes])
                t_.append(time.perf_counter())
                if n > 1 or not sharded:
                    self.rend.assemble_device_records_batch(p, k, self.gathered.data_ptr(), self.frames.data_ptr(), sp)
                    t_.append(time.perf_counter())
                    if host_copy[0]:
                        self.host_t[:k * frec].copy_(self.frames[:k * frec], non_blocking=True)
                else:
                    t_.append(time.perf_counter())
                    if host_copy[0]:
                        # emulated rank of a K-GPU run: its own records stand in for the gathered frames (same bytes per frame / K)
                        self.host_t[:k * record_bytes].copy_(self.records[:k * record_bytes], non_blocking=True)
                t_.append(time.perf_counter())
                for i_, name in enumerate(("render", "gather", "assemble", "copy")):  # slowest submission of each kind (host time)
                    submit_max[name] = max(submit_max[name], (t_[i_ + 1] - t_[i_]) * 1e3)

        def frame_rays(self):
            """Whole-frame ray count of the slot's last launch as it landed (host) or as the device holds it."""
            torch.cuda.synchronize()
            if sharded and n == 1:
                return int(self.records[trailer:record_bytes].view(torch.int64).item())
            if not host_copy[0]:  # device-resident run: bring the frames home to read the count
                if sharded or args.pixel_mode:
                    self.host_t[:frec].copy_(self.frames[:frec])
                else:
                    self.rend.render_batch_async(p, 1, self.host, 0, self.stream.cuda_stream)
                torch.cuda.synchronize()
            return self.host.rays(0)

        def local_rays(self):
            torch.cuda.synchronize()
            if not (sharded or args.pixel_mode):
                return self.frame_rays()
            return int(self.records[trailer:record_bytes].view(torch.int64).item())

    slots = [Slot() for _ in range(max(1, args.inflight))]
    rend = slots[0].rend
    counter = [0]

    def step():
        slots[(counter[0] // B) % len(slots)].step()
        counter[0] += 1

    def flush():
        for sl in slots:
            sl.launch()  # a partial last batch
        counter[0] = 0

    def fence():
        flush()
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(steps, with_events=False):
        """Time EXACTLY `steps` steps (frames) between two fences; max over ranks.  Returns (elapsed, submit, events)."""
        if with_events:
            for sl in slots:
                sl.rend.timing_begin(steps // (B * len(slots)) + 3)  # launches this slot will carry
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        flush()
        submit = time.perf_counter() - t0  # host time to enqueue the frames (must stay below `elapsed`)
        fence()
        elapsed = time.perf_counter() - t0
        ev = None
        if with_events:
            trace_ms, total_ms, launches = 0.0, 0.0, 0
            for sl in slots:
                a, b, c = sl.rend.timing_end()
                trace_ms, total_ms, launches = trace_ms + a, total_ms + b, launches + c
            ev = (trace_ms, total_ms, launches)
        if n > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, submit, ev

    # setup, not warm-up: frames through every slot so that its workspace (sample records, event ring, page-locked
    # frames) is allocated before anything is timed, whatever --warmup is.  Repeated until a whole pass goes through
    # without a stalled submission: the HIP runtime creates its hardware queues lazily, ~7 ms each, somewhere in the first
    # dozens of submissions on 16-20 streams (tools/submit_times.py: five such stalls in the first pass over 20 slots,
    # three in the second, none afterwards; a 20-step run that meets two of them is 16 ms late on a 17 ms job).
    def settle(at_least=2):
        """Passes over every slot until one goes through without a stalled submission; returns the number of passes.  Also run in front of
        every secondary measurement that uses other operations than the main one (a resolve launch and copies on streams that have only
        carried single-launch frames so far: their first uses stall as well — the exhaustive sweep once read 8.4 instead of 17.9 Grays/s)."""
        passes = 0
        for _ in range(8):
            worst = 0.0
            for _ in range(B * len(slots)):
                t_ = time.perf_counter()
                step()
                worst = max(worst, time.perf_counter() - t_)
            fence()
            passes += 1
            if n > 1:
                # every rank must run the SAME number of passes (each pass issues collectives): the verdict is the slowest rank's
                t = torch.tensor([worst], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                worst = float(t.item())
            if passes >= at_least and worst < 2e-3:
                break
        return passes

    setup_passes = settle()
    for _ in range(args.warmup):
        step()
    fence()
    slots[0].step()
    slots[0].launch()
    rays_per_step = slots[0].frame_rays()  # whole frame: the count that landed on the host
    if sharded and n == 1:
        rays_per_step = slots[0].local_rays()  # emulated rank: its share only

    for k_ in submit_max:
        submit_max[k_] = 0.0
    elapsed, submit, (trace_ms_sum, total_ms_sum, launches) = timed(args.steps, with_events=True)
    submit_max_timed = dict(submit_max)
    frames = args.steps
    value = rays_per_step * args.steps / elapsed / 1e6
    info = rend.launch_info()
    local_rays = slots[0].local_rays()  # this rank's rays: the work of ITS kernel launches

    check = None
    if args.check and not (sharded and n == 1):
        torch.cuda.synchronize()
        slots[0].step()
        slots[0].launch()
        assert slots[0].frame_rays() == rays_per_step  # (device-resident run: brings the frame home)
        got = slots[0].host.image(0).tobytes()
        ref = np.zeros((h, w, 3), np.uint8)
        ref_rays, _ = rend.render_into(r1.make_params(w, h, spp, args.seed, variant=args.variant), ref)
        check = bool(got == ref.tobytes() and rays_per_step == ref_rays)

    # ---- secondary measurements (rank 0's line only; every rank takes part in the timed regions) ----
    # N > 1 and a short run (the driver's 20 steps are a 2-4 ms timed region at N = 8: mostly ramp and drain): the same frames over
    # >= 200 steps beside it, so that the short region is not the only evidence of the N-GPU rate
    long_run = None
    if n > 1 and args.steps < 200 and not args.no_extras:
        steps3 = ((200 + B * len(slots) - 1) // (B * len(slots))) * (B * len(slots))
        for _ in range(B * len(slots)):
            step()
        e3, _, _ = timed(steps3)
        long_run = {"value": rays_per_step * steps3 / e3 / 1e6, "unit": "mrays/s", "steps": steps3, "ms_per_step": e3 / steps3 * 1e3,
                    "note": "the same job over a longer timed region (same frames in flight, same collectives): the steady-state rate of the N ranks"}
    resident = None
    if host_copy[0] and not args.no_extras:
        # the same frames left in HBM: what the copies to the host cost
        host_copy[0] = False
        for _ in range(B * len(slots)):
            step()
        steps2 = max(B * len(slots), args.steps // 2)
        e2, _, _ = timed(steps2)
        resident = {"value": rays_per_step * steps2 / e2 / 1e6, "unit": "mrays/s", "steps": steps2, "ms_per_step": e2 / steps2 * 1e3,
                    "mode": f"{len(slots) * B} frames in flight, images and ray counts left in HBM (round 2's headline mode)"}
        host_copy[0] = True

    # The same workload through the exhaustive sweep (the reference's algorithm: every ray against
    # every sphere), when `value` came from the box tree: reported beside it, never instead of it.
    sweep_line = None
    if n == 1 and args.variant == 0 and info["kernel"] == 4 and not sharded and info["spheres_active"] <= 1023 and not args.no_extras:
        p_main = p
        p = r1.make_params(w, h, spp, args.seed, shard=0, num_shards=1, variant=binding.VARIANT_PREFILTER)
        sweep_steps = max(B * len(slots), args.steps // 2, 100)  # (its own, longer region: a 20-step burst of this kernel reads 7 % lower, and `steps` says so)
        settle()
        sweep_elapsed, _, _ = timed(sweep_steps)
        sweep_rate = rays_per_step * sweep_steps / sweep_elapsed
        sweep_line = {"value": sweep_rate / 1e6, "unit": "mrays/s", "steps": sweep_steps,
                      "ms_per_step": sweep_elapsed / sweep_steps * 1e3,
                      "reference_equivalent": {
                          "bytes_per_ray": 16.0 * info["spheres_padded"], "flop_per_ray": 16.0 * info["spheres_padded"],
                          "tb_per_s": sweep_rate * 16.0 * info["spheres_padded"] / 1e12,
                          "tflop_per_s": sweep_rate * 16.0 * info["spheres_padded"] / 1e12,
                          "note": "SURVEY.md §8d's model of the REFERENCE's sweep (16 B and 16 flop per ray-sphere test x N_pad spheres "
                                  "per ray) at this kernel's ray rate.  Reference-equivalent, not executed, work: the kernel tests "
                                  "groups of <= 4 spheres from SGPRs, so these figures may exceed the hardware peaks (8 TB/s, 157.3 TFLOP/s)"},
                      "kernel": "grouped exhaustive sweep (R1_VARIANT_PREFILTER): every ray tested against every sphere group, "
                                "as the reference's Hitable::hit does; same pixels; pixels + count on the host like `value`"}
        p = p_main
        for _ in range(B):
            slots[0].step()  # leave launch_info / images describing the main kernel
        fence()
        info = rend.launch_info()

    # ---- what the timed kernel executes (measured, outside the timed region): one synchronous frame
    # through the diagnostic build of the same kernel (R1_VARIANT_*_STATS: same samples, plus counters)
    work = None
    if rank == 0 and (info["kernel"] == 4 or (info["kernel"] == 2 and info["spheres_active"] <= 1023)):
        try:
            work = measure_work(rend, p, info, binding)
        except Exception as e:  # diagnostics never fail the bench
            work = {"error": str(e)}

    if rank == 0:
        is_tree = info["kernel"] in (4, 5, 6)
        kernel_name = {1: "reference-form exhaustive sweep", 2: "grouped exhaustive sweep" + (" (LDS-tiled)" if info["spheres_active"] > 1023 else ""),
                       3: "grouped exhaustive sweep + counters", 4: "box tree (R1_VARIANT_BVH)", 5: "box tree + counters",
                       6: "wavefront: generate / intersect / shade kernels, box tree (comparison build)"}[info["kernel"]]
        kernel_s = trace_ms_sum / max(launches, 1) * 1e-3  # average launch duration, HIP events on the stream of each launch
        overlap = trace_ms_sum * 1e-3 / elapsed           # launches of different frames overlap (frames in flight)
        traffic, traffic_source, valu_instr, valu_source, valu_lanes, valu_busy, pmc_note = None, None, None, None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and info["kernel"] != 6:
            try:
                tj = json.load(open(tpath))
                # counters are only quoted for the build they were measured with: the file records the library's hash
                lib_now = lib_sha16(binding.lib_path())
                if tj.get("lib_sha16") != lib_now:
                    pmc_note = (f"profiles/pmc_traffic.json was measured with another build of librays1.so ({tj.get('lib_sha16')}, running {lib_now}): "
                                "not quoted; tools/profile_round.sh refreshes it")
                elif tj.get("workload") == f"{args.scene} {w}x{h}x{spp}" and n == 1 and not sharded and tj.get("kernel_variant") == info["kernel"]:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_traffic.json: {tj.get('source', 'rocprofv3 --pmc passes')} (not measured by this run)"
                    valu_instr = tj.get("valu_wave_instructions_per_launch")
                    valu_lanes = tj.get("valu_active_lane_fraction")
                    valu_busy = tj.get("valu_busy_fraction")
                    valu_source = f"{tj.get('valu_source', 'profiles/')} (not measured by this run)"
            except Exception:
                traffic = None
        flop_frame = work.get("flop_per_launch") if work else None  # measured on one synchronous frame = one frame's work
        flop = flop_frame * frames / max(launches, 1) if flop_frame else None  # per launch of the timed region (a batch of frames)
        agg = (flop_frame * frames / elapsed / 1e12) if flop_frame else None
        roofline = {
            # No dense contraction on this path (no MFMA) and the tables are cache resident, so neither of
            # the contract's two roofs binds: the roof is fp32 VECTOR issue.  `achieved` counts the flop the
            # launches EXECUTE in their hit tests (measured counts x flop per unit, below) — never the
            # reference-equivalent 16 B x N_pad model (that figure lives in `exhaustive_sweep`) — over the WALL time of
            # the timed region: with frames in flight the launches overlap, so a launch's own duration is not its share
            # of the machine (that per-launch figure is kept under `per_launch`).
            "bound": "valu", "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
            "achieved": agg,
            "frac": (agg / FP32_VECTOR_PEAK_TF) if agg else None,
            "traffic": traffic, "traffic_source": traffic_source, "traffic_note": pmc_note,
            "kernel": "r1_trace_kernel" if info["kernel"] != 6 else "r1_wf_generate + 51 x (r1_wf_intersect, r1_wf_shade)",
            "flop_per_frame": flop_frame, "flop_per_launch": flop, "frames_per_launch": frames / max(launches, 1),
            "work": work,
            "launch_overlap": overlap,
            "per_launch": {"kernel_ms": kernel_s * 1e3,  # HIP events around each launch, on its stream; rocprofv3's average agrees (profiles/)
                           "achieved": (flop / kernel_s / 1e12) if flop else None,
                           "frac": (flop / kernel_s / 1e12 / FP32_VECTOR_PEAK_TF) if flop else None,
                           "note": "flop per launch / that launch's own duration: under launch_overlap-deep overlap a launch only has a "
                                   "share of the chip, so this is a lower bound by that factor, not a utilisation"},
            # what binds the kernel: VALU wave-instructions issued (PMC) against one per two cycles per SIMD
            "valu_issue": {"wave_instructions_per_launch": valu_instr, "source": valu_source,
                           "active_lane_fraction": valu_lanes,  # SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes doing work per issued instruction
                           "busy_fraction": valu_busy,          # SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES per SIMD): measured with the frames in flight
                           "peak": VALU_ISSUE_PEAK_T, "unit": "T wave-instructions/s (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles)",
                           "achieved": (valu_instr * frames / elapsed / 1e12) if valu_instr else None,
                           "frac": (valu_instr * frames / elapsed / 1e12 / VALU_ISSUE_PEAK_T) if valu_instr else None,
                           "note": "the instructions of this kernel issue in 2.7 (fp32 add / mul / fma) to 4.4 cycles (min / max / compare / "
                                   "select / convert): profiles/r02/isa_issue_costs.txt, so ~0.6 of the 2-cycle peak is a busy pipe"},
            "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": (traffic * frames / elapsed / 1e9) if traffic else None,
                    "frac": (traffic * frames / elapsed / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "note": "measured HBM bytes per launch (PMC) x frames / wall time: the sample records and the image; sphere and node tables stay in L1/L2"},
            "note": "executed hit-test flop x frames / wall time / fp32 vector peak (157.3 TFLOP/s).  Low by construction: the count "
                    "leaves out shading, RNG, queue and control instructions, idle lanes of divergent traversals "
                    "(work.lane_utilisation) and the 4-cycle issue of VOP3 instructions; what binds is `valu_issue`.",
            # N > 1: the work is counted on rank 0's tiles and the peak is ONE GPU's — the figures describe rank 0's GPU, not the job
            "scope": "the one GPU" if n == 1 else f"rank 0's GPU (its tiles: 1/{n} of every frame) against one GPU's peak",
        }
        cfg = workload_config(args, info, n, rays_per_step)
        host_mb = (img_bytes + 8) / 1e6 if not (sharded and n == 1) else record_bytes / 1e6
        cfg.update({
            "parallelism": f"tile-split x{n}" + (" + one RCCL all-gather per frame (pixels + ray counts)" if n > 1 else ""),
            "value_mode": (f"{len(slots) * B} frames in flight = {len(slots)} launches (one stream + context each) x {B} frames per launch, "
                           f"scene resident in HBM; every frame ends with its pixels + ray count in page-locked HOST memory "
                           f"({host_mb:.2f} MB per frame) — " + ("stored there tile by tile by the trace launch itself (no resolve launch, no copy)"
                                                               if info.get("tiles_in_kernel") and not (sharded or args.pixel_mode) else
                                                               "copied on its launch's stream") +
                           ", and the timed region ends when all frames have landed (rayweek1.cpp:848 -> :891, pipelined)") if host_copy[0] else
                          f"{len(slots) * B} frames in flight ({len(slots)} launches x {B} frames), scene and images resident in HBM",
            "workgroups": info["blocks"], "threads_per_workgroup": info["threads_per_block"],
            "frames_in_flight": len(slots) * B, "launches_in_flight": len(slots), "frames_per_launch": B,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "host_submit_ms_per_step": submit / args.steps * 1e3, "setup_passes": setup_passes,
            "slowest_submission_ms": submit_max_timed if (sharded or args.pixel_mode) else None,
            "kernel": kernel_name})
        if is_tree:
            cfg["bvh"] = {"nodes": info["bvh_nodes"], "leaves": info["bvh_leaves"], "depth": info["bvh_depth"]}
        if sharded and n == 1:
            cfg["emulated_shards"] = shards
            cfg["local_rays_per_step"] = local_rays
        out = {
            "metric": f"mrays/s on '{args.scene}' scene {w}x{h}x{spp}spp",
            "value": value, "unit": "mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg,
            "roofline": roofline,
        }
        if long_run is not None:
            out["value_long_run"] = long_run
        if resident is not None:
            out["value_device_resident"] = resident
        if sweep_line is not None:
            out["exhaustive_sweep"] = sweep_line
        if check is not None:
            out["check"] = check
        if n == 1 and not sharded and not args.no_extras:
            # The survey's timer span (rayweek1.cpp:848 -> :891) for ONE frame the caller waits for, through r1_render() —
            # what the drop-in benchmark() prints: latency-mode kernels, one frame at a time.
            ph = r1.make_params(w, h, spp, args.seed, variant=args.variant)
            reps = max(5, min(args.steps, 50))

            def sync_frames(host):
                rend.render_into(ph, host)
                t1 = time.perf_counter()
                tot, dev_s = 0, 0.0
                for _ in range(reps):
                    r_, s_ = rend.render_into(ph, host)
                    tot, dev_s = tot + r_, dev_s + s_
                d2h = time.perf_counter() - t1
                return {"value": tot / d2h / 1e6, "unit": "mrays/s", "ms_per_step": d2h / reps * 1e3, "steps": reps, "device_ms_per_step": dev_s / reps * 1e3}

            # the pixel buffer as the drop-in program allocates it (r1_host_alloc: page-locked): the frame is ONE launch whose waves store
            # the tiles straight into it; and an ordinary (pageable) buffer, which receives one copy of the finished image
            hf_sync = binding.HostFrames(w, h, 1)
            out["value_dispatch_to_host"] = sync_frames(hf_sync.image(0))
            out["value_dispatch_to_host"]["span"] = ("r1_render(): launch -> pixels + ray count on the host (the reference's Timer span, rayweek1.cpp:848 -> :891), "
                                                     "one synchronous frame at a time, into a page-locked pixel buffer (r1_host_alloc, as rayweek1_hip allocates "
                                                     "`pixels`): one launch, the tiles land in the buffer as they complete, no copy")
            out["value_dispatch_to_host"]["pageable_buffer"] = sync_frames(np.zeros((h, w, 3), np.uint8))
            out["value_dispatch_to_host"]["pageable_buffer"]["span"] = "the same into ordinary memory: the launch + one copy of the finished image over PCIe"
            hf_sync.close()
        if n == 1 and not sharded and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_grid(slots[0].scene, w, h) if args.scene == "grid" else cpu_baseline(args.scene, w, h, spp)
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "mrays/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": f"failed: {e}"}
            if args.cpu_table:
                out["cpu_baseline"]["table"] = cpu_table(w, h, spp)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl.rend.close()
        sl.host.close()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.multi == "inproc":
        return run_inproc(args)
    world = os.environ.get("WORLD_SIZE")
    if world is None and args.gpus > 1:
        return self_launch(args)  # before torch / HIP are touched
    if world is not None and int(world) != args.gpus:
        args.gpus = int(world)  # the launcher decides how many ranks there are
    return run_ranks(args)


if __name__ == "__main__":
    sys.exit(main())
