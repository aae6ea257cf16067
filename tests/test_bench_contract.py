"""The bench line contract (no GPU): the committed line of the round (profiles/rNN/bench_line.json,
written by `python bench.py` on the GPU box) carries every field the driver and the judge read, and
bench.py's defaults are the contract's (N = 1, a K/W that finishes in minutes)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_line.json")))
    assert lines, "no committed bench line"
    d = json.loads(open(lines[-1]).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "mrays/s" and "large" in d["metric"] and "1200x800x10" in d["metric"]
    assert "large" in baseline["metric"] and "1200×800×10" in baseline["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == d["config"]["rays_per_step"] * 16.0 * 488  # 16 B x N_pad per ray
    assert r["traffic"] is None or 1e8 < r["traffic"] < 1e9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "mrays/s" and c["cores"] >= 1
    # the tree kernel's line also reports the exhaustive sweep beside it
    if "box tree" in d["config"].get("kernel", ""):
        assert d["exhaustive_sweep"]["value"] > 0


def test_bench_defaults_are_the_contracts():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    warmup = int(re.search(r'"--warmup", type=int, default=(\d+)', src).group(1))
    assert 10 <= steps <= 2000 and 1 <= warmup <= steps
