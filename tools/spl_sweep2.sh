# tools/spl_sweep2.sh — workgroups per frame (R1_SAMPLES_PER_LANE, tuning build) after DESIGN §4.13, long and short runs
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
for i in 1 2 3; do
for spl in 100 115 125 135; do
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "$get" "spl $spl steps 300"
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$get" "spl $spl steps 20"
done
done
for spl in 100 125; do
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --scene medium --steps 300 --warmup 20 2>/dev/null | python -c "$get" "spl $spl medium 300"
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --scene medium --steps 20 --warmup 5 2>/dev/null | python -c "$get" "spl $spl medium 20"
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --spp 250 --steps 48 --warmup 16 2>/dev/null | python -c "$get" "spl $spl spp250"
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --variant 2 --steps 200 --warmup 20 2>/dev/null | python -c "$get" "spl $spl sweep 200"
  R1_SAMPLES_PER_LANE=$spl python bench.py --lib $T --no-extras --no-cpu-baseline --emulate-shards 8 --rccl-selftest --steps 320 --warmup 32 2>/dev/null | python -c "$get" "spl $spl shards8"
done
