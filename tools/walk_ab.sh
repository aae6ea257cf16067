# tools/walk_ab.sh — the two walk experiments of round 4 against the product build, alternating on ONE box:
#   librays1_entry.so  make tuning EXTRA=-DR1_ENTRY=1 (copied)  per-tile entry nodes; knob R1_ENTRY_OFF=1: same code, table = the root's inner child
#   librays1_bvh4.so   make tuning EXTRA=-DR1_BVH4=1  (copied)  4-wide nodes
# (bench.py takes another build with --lib only; nothing is read from the environment)
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { # label lib extra-args...
  label=$1; lib=$2; shift 2
  python bench.py --lib $lib --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"
}
for i in 1 2 3; do
  run "product 300" $L/librays1.so --steps 300 --warmup 20
  run "entry   300" $L/librays1_entry.so --steps 300 --warmup 20
  R1_ENTRY_OFF=1 run "entryoff 300" $L/librays1_entry.so --steps 300 --warmup 20
  run "bvh4    300" $L/librays1_bvh4.so --steps 300 --warmup 20
  run "product 20" $L/librays1.so --steps 20 --warmup 5
  run "entry   20" $L/librays1_entry.so --steps 20 --warmup 5
  R1_ENTRY_OFF=1 run "entryoff 20" $L/librays1_entry.so --steps 20 --warmup 5
  run "bvh4    20" $L/librays1_bvh4.so --steps 20 --warmup 5
done
for i in 1 2; do
for lib in librays1.so librays1_entry.so librays1_bvh4.so; do
  run "$lib medium" $L/$lib --scene medium --steps 300 --warmup 20
  run "$lib spp250" $L/$lib --spp 250 --steps 48 --warmup 16
done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in librays1.so librays1_entry.so librays1_bvh4.so; do
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/walk_pmc_$lib -o p --output-format csv -- python3 $R/bench.py --lib $R/$L/$lib --no-cpu-baseline --no-extras --steps 3 --warmup 1 --inflight 1 > $R/gpurun_out/walk_pmc_$lib.log 2>&1 || echo "pmc $lib failed"
done
python3 - <<PY
import csv, glob, collections
for lib in ("librays1.so", "librays1_entry.so", "librays1_bvh4.so"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"$R/gpurun_out/walk_pmc_{lib}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "r1_trace_kernel<4, false, false, 0>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(lib, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
