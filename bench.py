#!/usr/bin/env python3
"""bench.py — mrays/s of the step13 hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full frame of the workload: every rank traces + resolves the tiles it owns
(tile t -> rank t % N, include/rays1.h r1_params.shard), then the dense per-rank tile blocks
(each with its 8-byte ray count appended) are all-gathered with RCCL (torch.distributed "nccl")
— ONE collective per frame — and the row-major image is assembled on every rank.  At N = 1 there is no collective.  The frame is
fixed (BASELINE: large scene, 1200x800x10 spp), so scaling is STRONG.  Inputs (sphere tables,
camera) are resident in HBM before the timed region; the image stays in HBM (the
PCIe-inclusive rate of the host-returning r1_render() is reported separately, never as
`value`).  Data is synthetic by construction: the scenes are code, not files.
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
        return {"value": rays / secs / 1e6, "unit": "mrays/s", "cores": rec[0]["threads"], "kind": "reference", "cpu_model": cpu_model(),
                "sample": f"{scene} {w}x{h}x{spp}: {len(rec)} frames through the reference's TileRenderScheduler/render_tile "
                          f"({os.path.basename(binary)}: flags of reference bench.py:175 with "
                          f"{'-march=x86-64-v3' if 'avx2' in os.path.basename(binary) else '-march=native of the build container'}), {secs:.1f} s"}
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


# flop per unit of the hit tests, counted from r1_kernels.hip (add / mul / compare / min / max = 1, fma = 2):
#   node visit   pad fma + 3 mul once, then per child box (bvh_box): 3 fma, 3 fma, 3 sub + 2 max, 3 add + 2 min, min + 2 cmp = 25
#                -> 5 + 2 x 25 = 55 (the round-1 form, which measured |m - o|^2 per box, was 2 x 32)
#   exact_offer  pass 1 of Hitable::hit for one sphere = SURVEY.md §8d's 16 flop (3 sub, mul + 2 fma, mul + 2 fma, sub, mul + sub)
#   group test   7 fma + 1 sub of the prefilter (sweep_prefilter)                                                       = 15
FLOP_NODE, FLOP_SPHERE, FLOP_GROUP = 55.0, 16.0, 15.0


def measure_work(rend, p, info, binding):
    """Counts what one launch of the timed kernel executes: a synchronous frame through the
    diagnostic build of the same kernel (same samples; tools/bvh_stats.py, tools/kernel_stats.py)."""
    import ctypes as C
    import numpy as np
    stats_variant = binding.VARIANT_BVH_STATS if info["kernel"] == 4 else binding.VARIANT_STATS
    q = binding.Params.from_buffer_copy(p)
    q.variant = stats_variant
    host = np.zeros((p.height, p.width, 3), np.uint8)
    rays, _ = rend.render_into(q, host)
    st = rend.last_stats()
    it = max(st["wave_iterations"], 1)
    if info["kernel"] == 4:
        # slot "cycles_pass1" of the tree build: sphere-pair tests (low 32 bits) | leaf trips summed over lanes << 32
        visits, pairs, leaf_lane_trips = st["candidates"], st["cycles_pass1"] & 0xFFFFFFFF, st["cycles_pass1"] >> 32
        return {"source": "R1_VARIANT_BVH_STATS frame (same samples as the timed kernel)", "rays_per_launch": rays,
                "node_visits_per_ray": visits / rays, "sphere_pair_tests_per_ray": pairs / rays,
                "flop_per_node_visit": FLOP_NODE, "flop_per_sphere_pair_test": 2 * FLOP_SPHERE,
                "flop_per_launch": visits * FLOP_NODE + pairs * 2 * FLOP_SPHERE,
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


def main():
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
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight, each on its own stream + context workspace: later frames' workgroups fill "
                         "the CUs that a frame's last long bounce chains leave idle (1 = one frame at a time; "
                         "0 = default: 20 on one GPU, 16 per rank when RCCL needs hardware queues of its own)")
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="tuning aid: render only shard 0 of K on one GPU, no collective (per-rank load of a K-GPU run)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region compare the gathered + assembled image and ray count with an unsharded render")
    ap.add_argument("--pixel-mode", action="store_true",
                    help="r1_set_pixel_mode: lanes own pixels, 3 B/pixel written, no per-sample workspace, no resolve launch (~10 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-table", action="store_true",
                    help="also time the reference's step13 (all threads / 1 thread) on the three scenes and the step1 port "
                         "on the host cores (cpu_baseline.table; adds ~30 s)")
    args = ap.parse_args()

    # Frames in flight only overlap when their streams sit on different hardware queues; the
    # ROCm default is 4 (measured: 16 queues + 16 frames in flight reach 98 % / 91 % of the ideal
    # per-rank frame time at 1/2/4 and 8 shards, 4 queues 78 % / 66 %).  Must be set before HIP starts.
    # One queue per frame in flight: streams that share a queue serialise, and spare queues hurt too
    # (measured: 16 in flight on 16 queues 1.18 ms per frame; 12 on 16: 1.77; 12 on 12: 1.20; 24 on 24: 1.53).
    # 20 in flight on 20 queues: the driver's short run (20 steps) 26.3 -> 27.1 Grays/s, a long run unchanged (29.6); 24 on 24
    # collapses (10-24 Grays/s, the process runs out of hardware queues), so ranks that also run RCCL stay at 16.
    if args.inflight <= 0:
        args.inflight = 20 if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.emulate_shards <= 1 else 16
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(1, min(args.inflight, 24))))
    import torch
    import rays1bench_amd as r1
    from rays1bench_amd import binding, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        n = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU fallback)")
    if os.environ.get("R1_BENCH_DEVICE"):  # rehearsal: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["R1_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if n > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend != "nccl":
            dist.init_process_group(args.backend, rank=rank, world_size=n)
        else:
            try:
                dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", local_rank))
            except TypeError:  # older torch: no device_id keyword
                dist.init_process_group("nccl", rank=rank, world_size=n)

    w, h, spp = args.width, args.height, args.spp
    dev = torch.device("cuda", local_rank)
    p = r1.make_params(w, h, spp, args.seed, shard=rank, num_shards=n, variant=args.variant)
    if args.emulate_shards > 1 and n == 1:
        p = r1.make_params(w, h, spp, args.seed, shard=0, num_shards=args.emulate_shards, variant=args.variant)
    block_bytes = binding.shard_block_bytes(p)

    shards = max(n, args.emulate_shards, 1)
    record_bytes = block_bytes + sharding.RECORD_TRAILER  # tile block + uint64 ray count: one all-gather per frame

    class Slot:
        """One frame in flight: its own context (stream-ordered workspace), buffers and stream."""

        def __init__(self):
            self.rend = r1.Renderer(local_rank)
            gw, gh = (int(v) for v in args.grid.split("x")) if args.scene == "grid" else (0, 0)
            self.scene = r1.Scene(SCENE_KIND[args.scene], w, h, gw, gh)
            self.rend.set_scene(self.scene)
            if args.pixel_mode:
                self.rend.set_pixel_mode(True)
            self.record = torch.zeros(record_bytes, dtype=torch.uint8, device=dev)
            self.gathered = torch.zeros(shards * record_bytes, dtype=torch.uint8, device=dev) if shards > 1 else self.record
            self.image = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
            self.stream = torch.cuda.Stream(device=dev)

        def step(self):
            with torch.cuda.stream(self.stream):
                sp = self.stream.cuda_stream
                self.rend.render_shard_device(p, self.record.data_ptr(), self.record.data_ptr() + block_bytes, sp)
                if n > 1:
                    sharding.gather_records(dist, self.record, self.gathered)
                self.rend.assemble_device_strided(p, self.gathered.data_ptr(), record_bytes, self.image.data_ptr(), sp)

        def local_rays(self):
            return int(self.record[block_bytes:].view(torch.int64).item())

        def frame_rays(self):
            return sharding.total_rays(self.gathered, shards) if n > 1 else self.local_rays()

    slots = [Slot() for _ in range(max(1, args.inflight))]
    rend = slots[0].rend
    counter = [0]

    def step():
        slots[counter[0] % len(slots)].step()
        counter[0] += 1

    def fence():
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # setup, not warm-up: one frame through every slot so that its workspace (sample records,
    # event ring) is allocated before anything is timed, whatever --warmup is
    for sl in slots:
        sl.step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    slots[0].step()
    torch.cuda.synchronize()
    rays_per_step = slots[0].frame_rays()  # whole frame: the sum of the gathered trailers

    for sl in slots:
        sl.rend.timing_begin(args.steps // len(slots) + 2)  # frames this slot will carry
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    submit = time.perf_counter() - t0  # host time to enqueue the frames (must stay below `elapsed`)
    fence()
    elapsed = time.perf_counter() - t0
    trace_ms_sum, total_ms_sum, frames = 0.0, 0.0, 0
    for sl in slots:
        a, b, c = sl.rend.timing_end()
        trace_ms_sum, total_ms_sum, frames = trace_ms_sum + a, total_ms_sum + b, frames + c
    if n > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = rays_per_step * args.steps / elapsed / 1e6
    info = rend.launch_info()

    # The same workload through the exhaustive sweep (the reference's algorithm: every ray against
    # every sphere), when `value` came from the box tree: reported beside it, never instead of it.
    sweep_line = None
    if n == 1 and args.variant == 0 and info["kernel"] == 4 and args.emulate_shards <= 1 and info["spheres_active"] <= 1023:
        p_main = p
        p = r1.make_params(w, h, spp, args.seed, shard=0, num_shards=1, variant=binding.VARIANT_PREFILTER)
        sweep_steps = max(len(slots), args.steps // 2)
        for _ in range(len(slots)):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(sweep_steps):
            step()
        fence()
        sweep_elapsed = time.perf_counter() - t1
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
                                "as the reference's Hitable::hit does; same pixels"}
        p = p_main
        slots[0].step()  # leave launch_info / images describing the main kernel
        torch.cuda.synchronize()
        info = rend.launch_info()

    # local rays of this rank for the roofline of ITS kernel launches
    local_rays = slots[0].local_rays()

    check = None
    if args.check:
        import numpy as np
        torch.cuda.synchronize()
        got = slots[0].image.cpu().numpy()
        got_rays = rays_per_step
        ref = np.zeros((h, w, 3), np.uint8)
        ref_rays, _ = rend.render_into(r1.make_params(w, h, spp, args.seed, variant=args.variant), ref)
        check = bool(got.tobytes() == ref.tobytes() and got_rays == ref_rays)

    # ---- what the timed kernel executes (measured, outside the timed region): one synchronous frame
    # through the diagnostic build of the same kernel (R1_VARIANT_*_STATS: same samples, plus counters)
    work = None
    if rank == 0 and (info["kernel"] == 4 or (info["kernel"] == 2 and info["spheres_active"] <= 1023)):
        try:
            work = measure_work(rend, p, info, binding)
        except Exception as e:  # diagnostics never fail the bench
            work = {"error": str(e)}

    if rank == 0:
        n_pad = info["spheres_padded"]
        is_tree = info["kernel"] in (4, 5, 6)
        kernel_name = {1: "reference-form exhaustive sweep", 2: "grouped exhaustive sweep" + (" (LDS-tiled)" if info["spheres_active"] > 1023 else ""),
                       3: "grouped exhaustive sweep + counters", 4: "box tree (R1_VARIANT_BVH)", 5: "box tree + counters",
                       6: "wavefront: generate / intersect / shade kernels, box tree (comparison build)"}[info["kernel"]]
        kernel_s = trace_ms_sum / max(frames, 1) * 1e-3  # average launch duration, HIP events on the stream of each launch
        overlap = trace_ms_sum * 1e-3 / elapsed           # launches of different frames overlap (frames in flight)
        traffic, traffic_source, valu_instr, valu_source, valu_lanes = None, None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and info["kernel"] != 6:
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"{args.scene} {w}x{h}x{spp}" and n == 1 and tj.get("kernel_variant") == info["kernel"]:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_traffic.json: {tj.get('source', 'rocprofv3 --pmc passes')} (not measured by this run)"
                    valu_instr = tj.get("valu_wave_instructions_per_launch")
                    valu_lanes = tj.get("valu_active_lane_fraction")
                    valu_source = f"{tj.get('valu_source', 'profiles/')} (not measured by this run)"
            except Exception:
                traffic = None
        flop = work.get("flop_per_launch") if work else None
        roofline = {
            # No dense contraction on this path (no MFMA) and the tables are cache resident, so neither of
            # the contract's two roofs binds: the roof is fp32 VECTOR issue.  `achieved` counts the flop the
            # launch EXECUTES in its hit tests (measured counts x flop per unit, below) — never the
            # reference-equivalent 16 B x N_pad model (that figure lives in `exhaustive_sweep`).
            "bound": "valu", "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
            "achieved": (flop / kernel_s / 1e12) if flop else None,
            "frac": (flop / kernel_s / 1e12 / FP32_VECTOR_PEAK_TF) if flop else None,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "r1_trace_kernel" if info["kernel"] != 6 else "r1_wf_generate + 51 x (r1_wf_intersect, r1_wf_shade)",
            "kernel_ms": kernel_s * 1e3,
            "flop_per_launch": flop,
            "work": work,
            # with several frames in flight the launches overlap: `achieved` is per launch as the contract
            # defines it (flop per launch / average launch duration); x launch_overlap = what the chip sustains
            "launch_overlap": overlap,
            "achieved_aggregate": (flop * frames / elapsed / 1e12) if flop else None,
            "frac_aggregate": (flop * frames / elapsed / 1e12 / FP32_VECTOR_PEAK_TF) if flop else None,
            # what binds the kernel: VALU wave-instructions issued (PMC) against one per two cycles per SIMD
            "valu_issue": {"wave_instructions_per_launch": valu_instr, "source": valu_source,
                           "active_lane_fraction": valu_lanes,  # SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes doing work per issued instruction
                           "peak": 1024 * 2.4e9 / 2 / 1e12, "unit": "T wave-instructions/s (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles)",
                           "achieved_aggregate": (valu_instr * frames / elapsed / 1e12) if valu_instr else None,
                           "frac_aggregate": (valu_instr * frames / elapsed / 1e12 / (1024 * 2.4e9 / 2 / 1e12)) if valu_instr else None,
                           "note": "the instructions of this kernel issue in 2.7 (fp32 add / mul / fma) to 4.4 cycles (min / max / compare / "
                                   "select / convert): profiles/r02/isa_issue_costs.txt, so ~0.6 of the 2-cycle peak is a busy pipe"},
            "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": (traffic / kernel_s / 1e9) if traffic else None,
                    "achieved_aggregate": (traffic * frames / elapsed / 1e9) if traffic else None,
                    "frac_aggregate": (traffic * frames / elapsed / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "note": "measured HBM bytes per launch (PMC) / time: the sample records and the image; sphere and node tables stay in L1/L2"},
            "note": "executed hit-test flop / launch duration / fp32 vector peak (157.3 TFLOP/s).  Low by construction: the count "
                    "leaves out shading, RNG, queue and control instructions, idle lanes of divergent traversals "
                    "(work.lane_utilisation) and the 4-cycle issue of VOP3 instructions; VALU issue-busy cycles are in "
                    "profiles/ (DESIGN.md §7).",
        }
        out = {
            "metric": f"mrays/s on '{args.scene}' scene {w}x{h}x{spp}spp",
            "value": value, "unit": "mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} scene ({info['spheres_active']} spheres, N_pad {n_pad}), {w}x{h}, {spp} spp, "
                                   f"max 50 bounces, seed {args.seed}",
                       "rays_per_step": rays_per_step, "tiles": "32x32, tile t -> rank t % N",
                       "parallelism": f"tile-split x{n}" + (" + one RCCL all-gather per frame (pixels + ray counts)" if n > 1 else ""),
                       "value_mode": f"{len(slots)} frames in flight (one stream + context each), scene and image resident in HBM",
                       "workgroups": info["blocks"], "threads_per_workgroup": info["threads_per_block"],
                       "frames_in_flight": len(slots), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "host_submit_ms_per_step": submit / args.steps * 1e3,
                       "kernel": kernel_name,
                       **({"bvh": {"nodes": info["bvh_nodes"], "leaves": info["bvh_leaves"], "depth": info["bvh_depth"]}} if is_tree else {})},
            "roofline": roofline,
        }
        if sweep_line is not None:
            out["exhaustive_sweep"] = sweep_line
        if check is not None:
            out["check"] = check
        if n == 1:
            # The survey's timer span (rayweek1.cpp:848 -> :891): dispatch -> pixels + ray count on the HOST,
            # one synchronous frame at a time through r1_render() — what the drop-in benchmark() prints.
            import numpy as np
            host = np.zeros((h, w, 3), np.uint8)
            ph = r1.make_params(w, h, spp, args.seed, variant=args.variant)
            rend.render_into(ph, host)
            reps = max(5, min(args.steps, 50))
            t1 = time.perf_counter()
            tot, dev_s = 0, 0.0
            for _ in range(reps):
                r_, s_ = rend.render_into(ph, host)
                tot, dev_s = tot + r_, dev_s + s_
            d2h = time.perf_counter() - t1
            out["value_dispatch_to_host"] = {
                "value": tot / d2h / 1e6, "unit": "mrays/s", "ms_per_step": d2h / reps * 1e3, "steps": reps,
                "device_ms_per_step": dev_s / reps * 1e3,
                "span": "r1_render(): launch -> pixels + ray count on the host (the reference's Timer span, rayweek1.cpp:848 -> :891), "
                        "one frame at a time, PCIe copy included"}
            if not args.no_cpu_baseline:
                try:
                    out["cpu_baseline"] = cpu_baseline_grid(slots[0].scene, w, h) if args.scene == "grid" else cpu_baseline(args.scene, w, h, spp)
                except Exception as e:  # the baseline is reported, never required
                    out["cpu_baseline"] = {"value": None, "unit": "mrays/s", "cores": os.cpu_count(), "kind": "reference",
                                           "sample": f"failed: {e}"}
                if args.cpu_table:
                    out["cpu_baseline"]["table"] = cpu_table(w, h, spp)
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()
    for sl in slots:
        sl.rend.close()


if __name__ == "__main__":
    main()
