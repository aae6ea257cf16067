"""The bench line contract (no GPU): the committed line of the round (profiles/rNN/bench_line.json,
written by `python bench.py` on the GPU box) carries every field the driver and the judge read, no
field named `frac` exceeds 1, every roofline number can be recomputed from the other fields of the
line, and bench.py's defaults are the contract's (N = 1, a K/W that finishes in minutes)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_line.json")))
    assert lines, "no committed bench line"
    return lines[-1], json.loads(open(lines[-1]).read())


def fracs(node, path=""):
    if isinstance(node, dict):
        for k, v in node.items():
            if k.startswith("frac") and v is not None:
                yield path + "/" + k, v
            yield from fracs(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from fracs(v, f"{path}[{i}]")


def test_committed_bench_line_has_the_contract_fields():
    path, d = latest_line()
    assert os.path.basename(os.path.dirname(path)) >= "r03", "the round's line has not been committed"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "value_dispatch_to_host", "value_device_resident"):
        assert k in d, k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "mrays/s" and "large" in d["metric"] and "1200x800x10" in d["metric"]
    assert "large" in baseline["metric"] and "1200×800×10" in baseline["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    # `value` times the survey's span — every frame in flight ends with its pixels + count on the HOST
    mode = d["config"]["value_mode"]
    assert "in flight" in mode and "page-locked HOST memory" in mode and "rayweek1.cpp:848 -> :891" in mode
    r = d["value_device_resident"]
    assert r["unit"] == "mrays/s" and "left in HBM" in r["mode"] and 0.8 * d["value"] < r["value"] < 1.25 * d["value"]


def test_roofline_describes_the_timed_kernel_and_no_frac_exceeds_one():
    _, d = latest_line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "flop_per_launch", "flop_per_frame", "work", "launch_overlap",
              "per_launch", "valu_issue", "hbm"):
        assert k in r, k
    for where, v in fracs(d):
        assert 0 <= v <= 1, (where, v)
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    # `achieved` / `frac` are the AGGREGATE figures: executed flop of the frames of the timed region / its wall time
    assert abs(r["achieved"] - r["flop_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the per-launch figure (flop per launch / that launch's own duration under overlap) lives in a sub-record
    pl = r["per_launch"]
    assert abs(pl["achieved"] - r["flop_per_launch"] / (pl["kernel_ms"] * 1e-3) / 1e12) < 1e-9 * pl["achieved"]
    assert abs(pl["frac"] - pl["achieved"] / r["peak"]) < 1e-12 and pl["frac"] <= r["frac"]
    assert abs(r["launch_overlap"] - pl["kernel_ms"] / (d["ms_per_step"] * r["frames_per_launch"])) < 0.05 * r["launch_overlap"]
    assert abs(r["flop_per_launch"] - r["flop_per_frame"] * r["frames_per_launch"]) < 1e-6 * r["flop_per_launch"]
    # executed work: measured counts x the stated flop per unit
    w = r["work"]
    assert w["rays_per_launch"] == d["config"]["rays_per_step"]
    if "box tree" in d["config"]["kernel"]:
        flop = w["rays_per_launch"] * (w["node_visits_per_ray"] * w["flop_per_node_visit"] + w["root_steps_per_ray"] * w["flop_per_root_step"]
                                       + w["sphere_pair_tests_per_ray"] * w["flop_per_sphere_pair_test"])
        assert w["flop_per_node_visit"] == 50 and w["flop_per_sphere_pair_test"] == 32 and w["flop_per_root_step"] == 25
        assert 0 <= w["root_steps_per_ray"] <= 1
        assert 0 < w["lane_utilisation"]["node_loop"] <= 1 and 0 < w["lane_utilisation"]["leaf_loop"] <= 1
    else:
        flop = w["rays_per_launch"] * w["group_tests_per_ray"] * w["flop_per_group_test"] + w["rays_per_launch"] * w["exact_slots_per_ray"] * w["flop_per_exact_slot"]
    assert abs(flop - r["flop_per_frame"]) < 1e-6 * flop
    assert flop < 16.0 * 488 * w["rays_per_launch"]  # far below the reference-equivalent count: that model is not in `roofline`
    # HBM traffic and VALU issue: either absent or labelled with where they were measured
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    if r["traffic"] is not None:
        assert "profiles/" in r["traffic_source"] and os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json"))
        assert r["hbm"]["peak"] == 8000.0 and 0 < r["hbm"]["frac"] < 1
        assert abs(r["hbm"]["achieved"] - r["traffic"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["hbm"]["achieved"]
        v = r["valu_issue"]
        assert "profiles/r0" in v["source"] and 0 < v["frac"] <= 1 and 0 < v["active_lane_fraction"] <= 1
        assert abs(v["achieved"] - v["wave_instructions_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * v["achieved"]


def test_dispatch_to_host_cpu_baseline_and_sweep_subrecord():
    _, d = latest_line()
    v = d["value_dispatch_to_host"]
    assert v["unit"] == "mrays/s" and 0 < v["value"] <= d["value"] * 1.05 and "rayweek1.cpp:848" in v["span"] and "synchronous" in v["span"]
    assert abs(v["value"] - d["config"]["rays_per_step"] / v["ms_per_step"] / 1e3) < 1e-6 * v["value"]
    assert v["device_ms_per_step"] <= v["ms_per_step"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "note"):
        assert k in c, k
    assert "thread" in c["note"]  # say next to the number what it mostly measures at this frame size
    assert c["kind"] in ("reference", "port") and c["unit"] == "mrays/s" and c["cores"] >= 1 and len(c["cpu_model"]) > 3
    if "box tree" in d["config"].get("kernel", ""):
        e = d["exhaustive_sweep"]
        assert e["value"] > 0
        q = e["reference_equivalent"]  # the survey's 16 B x N_pad model lives here, labelled, and carries no `frac`
        assert q["bytes_per_ray"] == 16 * 488 and "Reference-equivalent" in q["note"]
        assert abs(q["tb_per_s"] - e["value"] * 1e6 * q["bytes_per_ray"] / 1e12) < 1e-6 * q["tb_per_s"]


def test_bench_defaults_are_the_contracts():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    warmup = int(re.search(r'"--warmup", type=int, default=(\d+)', src).group(1))
    assert 10 <= steps <= 2000 and 1 <= warmup <= steps


def test_bare_gpus_n_launches_fresh_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` (no launcher): the parent starts torch.distributed.run with one rank per GPU on
    127.0.0.1 and relays the exit code — without importing torch or librays1 itself (it used to exit)."""
    import subprocess
    import sys
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['bench.py', '--gpus', '4', '--steps', '7']\n"
                            "import bench, subprocess\n"
                            "seen = {}\n"
                            "def fake(cmd, **kw):\n"
                            "    seen['cmd'] = cmd; seen['env'] = kw.get('env', {})\n"
                            "    class R: returncode = 17\n"
                            "    return R()\n"
                            "subprocess.run = fake\n"
                            "rc = bench.main()\n"
                            "assert rc == 17, rc\n"
                            "c = seen['cmd']\n"
                            "assert c[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in c and c[c.index('--nproc-per-node') + 1] == '4'\n"
                            "assert c[c.index('--master-addr') + 1] == '127.0.0.1' and c[-4:] == ['--gpus', '4', '--steps', '7']\n"
                            "assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
                            "assert 'torch' not in sys.modules and 'rays1bench_amd' not in sys.modules\n"
                            "print('ok')"], capture_output=True, text=True, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert probe.returncode == 0 and probe.stdout.strip() == "ok", probe.stderr[-1500:]


def test_hardware_queue_budget_is_explicit_and_below_the_cliff():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ["GPU_MAX_HW_QUEUES"] = str(q)' in src and "setdefault(\"GPU_MAX_HW_QUEUES\"" not in src
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    assert bench.INFLIGHT_SINGLE <= bench.QUEUES_SINGLE < bench.QUEUES_CLIFF and bench.INFLIGHT_RANK < bench.QUEUES_RANK < bench.QUEUES_CLIFF
