# tools/walk_ab.sh — the two walk experiments of round 4 against the product build, alternating on ONE box.  Build first:
#   make -C rays1bench_amd/csrc tuning                     && cp rays1bench_amd/lib/librays1_tuning.so rays1bench_amd/lib/librays1_plain.so   (control)
#   make -C rays1bench_amd/csrc tuning EXTRA=-DR1_ENTRY=1  && cp rays1bench_amd/lib/librays1_tuning.so rays1bench_amd/lib/librays1_entry.so   (per-tile entry nodes)
#   make -C rays1bench_amd/csrc tuning EXTRA=-DR1_BVH4=1   && cp rays1bench_amd/lib/librays1_tuning.so rays1bench_amd/lib/librays1_bvh4.so    (4-wide nodes)
# knobs of the entry build: R1_ENTRY_OFF=1 (same code, the table holds the root's inner child), R1_ENTRY_LDS=0 (table in global memory).
# bench.py takes another build with --lib ONLY (nothing is read from the environment).
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { # label library bench-arguments...
  local label=$1 so=$2; shift 2
  python bench.py --lib $so --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"
}
for i in 1 2 3; do
  for steps in "300 20" "20 5"; do
    set -- $steps
    run "product $1" $L/librays1.so --steps $1 --warmup $2
    run "plain-tuning $1" $L/librays1_plain.so --steps $1 --warmup $2
    run "entry(lds) $1" $L/librays1_entry.so --steps $1 --warmup $2
    R1_ENTRY_OFF=1 run "entry(lds, table off) $1" $L/librays1_entry.so --steps $1 --warmup $2
    R1_ENTRY_LDS=0 run "entry(global) $1" $L/librays1_entry.so --steps $1 --warmup $2
    R1_ENTRY_LDS=0 R1_ENTRY_OFF=1 run "entry(global, table off) $1" $L/librays1_entry.so --steps $1 --warmup $2
    run "bvh4 $1" $L/librays1_bvh4.so --steps $1 --warmup $2
  done
done
for so in librays1.so librays1_entry.so librays1_bvh4.so; do
  run "$so medium" $L/$so --scene medium --steps 300 --warmup 20
  run "$so spp250" $L/$so --spp 250 --steps 48 --warmup 16
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in librays1.so librays1_entry.so librays1_bvh4.so; do
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/walk_pmc_$so -o p --output-format csv -- python3 $R/bench.py --lib $R/$L/$so --no-cpu-baseline --no-extras --steps 3 --warmup 1 --inflight 1 > $R/gpurun_out/walk_pmc_$so.log 2>&1 || echo "pmc $so failed"
done
python3 - <<PY
import csv, glob, collections
for so in ("librays1.so", "librays1_entry.so", "librays1_bvh4.so"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"$R/gpurun_out/walk_pmc_{so}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "r1_trace_kernel<4, false, false, 0>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(so, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
