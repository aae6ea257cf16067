# tools/fit_grid.sh — workgroups per frame chosen so that ALL frames in flight are resident at once (1792 slots / 20 frames = 89)
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
for i in 1 2; do
for cfg in "125 128" "200 64" "300 64" "421 64" "600 32"; do
  set -- $cfg
  R1_SAMPLES_PER_LANE=$1 R1_MIN_BLOCKS=$2 python bench.py --lib $T --no-extras --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "$get" "spl $1 minb $2 steps 300"
  R1_SAMPLES_PER_LANE=$1 R1_MIN_BLOCKS=$2 python bench.py --lib $T --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$get" "spl $1 minb $2 steps 20"
done
done
