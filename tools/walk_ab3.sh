# tools/walk_ab3.sh — second half of tools/walk_ab.sh: the other configurations and the VALU counters of the three builds
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { label=$1; lib=$2; shift 2; python bench.py --lib $lib --no-extras --no-cpu-baseline "$@" 2>gpurun_out/walk_err.txt | python -c "$get" "$label" || tail -5 gpurun_out/walk_err.txt; }
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
