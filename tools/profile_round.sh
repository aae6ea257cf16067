#!/bin/bash
# tools/profile_round.sh TAG — the round's evidence set, run on the GPU box; results under
# gpurun_out/prof_TAG/ (copy what should be judged into profiles/rNN/).
#   1. rocprofv3 --kernel-trace --stats of the default bench command (16 frames in flight)
#   2. the same with --inflight 1 (one launch at a time: a launch's duration = its share of the chip)
#   3. HBM traffic of the trace kernel: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
#   4. the bench line itself (with cpu_baseline) and the emulated per-rank load of N-GPU runs
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/k16 -o k16 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/k16.log 2>&1 || echo "k16 failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/k1 -o k1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --inflight 1 > $OUT/k1.log 2>&1 || echo "k1 failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --inflight 1 > $OUT/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
cd $R
timeout -k 10 300 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || echo "bench failed"
for k in 2 4 8; do
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --emulate-shards $k 2>/dev/null | tail -1 > $OUT/emulate_shards_$k.json || echo "emulate $k failed"
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
for tag in ("k16", "k1"):
    for f in glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True):
        print(tag, open(f).read().splitlines()[1][:160])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k, "mean per dispatch", sum(v) / len(v), "n", len(v))
d = json.loads(open(f"{out}/bench_line.json").read().strip().splitlines()[-1])
print("bench value", d["value"], "ms", d["ms_per_step"], "cpu", d.get("cpu_baseline", {}).get("value"))
for k in (2, 4, 8):
    e = json.loads(open(f"{out}/emulate_shards_{k}.json").read())
    print("emulated shards", k, "ms/step", e["ms_per_step"])
PY
