# tools/retune_r04.sh — knobs of the throughput path re-checked after DESIGN §4.13 (tuning builds; one box).  Earlier in the round
# (same script, other knobs): R1_CHUNK 256 (default) / 512 / 2048 -> 37.0 / 36.8 / 36.7 Grays/s; R1_CARRY_DIV 4 / 6 (default) / 8 ->
# 37.0 / 37.0 / 36.7; frames in flight 14 / 16 / 20 (default) / 24 -> 36.7 / 36.9 / 37.0 / 36.8 (20 steps: 31.7 / 32.6 / 33.4 / 33.4)
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2 3; do
  for n in librays1 librays1_spare0 librays1_smin28 librays1_smin52; do
    run "$n 300" $L/$n.so --steps 300 --warmup 20
    run "$n 20" $L/$n.so --steps 20 --warmup 5
  done
done
