# tools/fresh_ab.sh — kernel arguments re-read from the kernarg segment where they are used (R1_FRESH) against keeping them in (spilled) SGPRs
#   build first: make -C rays1bench_amd/csrc tuning EXTRA=-DR1_FRESH=0 && cp rays1bench_amd/lib/librays1_tuning.so rays1bench_amd/lib/librays1_nofresh.so
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/fresh_tests.log 2>&1 || { tail -30 gpurun_out/fresh_tests.log; exit 1; }
tail -2 gpurun_out/fresh_tests.log
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"], "sync", (d.get("value_dispatch_to_host") or {}).get("value"))'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2 3; do
  for so in librays1.so librays1_nofresh.so; do
    run "$so 300" $L/$so --steps 300 --warmup 20 --no-extras
    run "$so 20" $L/$so --steps 20 --warmup 5 --no-extras
  done
done
for so in librays1.so librays1_nofresh.so; do
  run "$so medium" $L/$so --scene medium --steps 300 --warmup 20 --no-extras
  run "$so spp250" $L/$so --spp 250 --steps 48 --warmup 16 --no-extras
  run "$so sweep" $L/$so --variant 2 --steps 200 --warmup 20 --no-extras
  run "$so config5" $L/$so --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 --no-extras
  run "$so full-line" $L/$so
done
