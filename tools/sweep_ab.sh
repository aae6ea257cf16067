# tools/sweep_ab.sh — the exhaustive sweep: A/B of one build against the product, alternating on ONE box
#   usage: tools/sweep_ab.sh rays1bench_amd/lib/<other>.so    (e.g. make tuning EXTRA=-DR1_SWEEP_PAIRS2=0, copied)
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
OTHER=${1:-$L/librays1_pairs1.so}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or prefilter or reference or big_scene or config5 or sphere_count" > gpurun_out/sweep_tests.log 2>&1 || { tail -30 gpurun_out/sweep_tests.log; exit 1; }
tail -2 gpurun_out/sweep_tests.log
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --variant 2 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2 3; do
  for so in $L/librays1.so $OTHER; do
    run "$so 300" $so --steps 300 --warmup 20
    run "$so 100" $so --steps 100 --warmup 20
  done
done
run "product medium" $L/librays1.so --scene medium --steps 300 --warmup 20
run "other medium" $OTHER --scene medium --steps 300 --warmup 20
