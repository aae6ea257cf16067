# tools/retune_r04.sh — knobs of the throughput path re-checked after DESIGN §4.13 (tuning build; one box)
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib; T=$L/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2; do
  run "base 300" $T --steps 300 --warmup 20
  run "base 20" $T --steps 20 --warmup 5
  for c in 512 2048; do
    R1_CHUNK=$c run "chunk $c 300" $T --steps 300 --warmup 20
    R1_CHUNK=$c run "chunk $c 20" $T --steps 20 --warmup 5
  done
  for d in 4 8; do
    run "carry_div $d 300" $L/librays1_carry$d.so --steps 300 --warmup 20
    run "carry_div $d 20" $L/librays1_carry$d.so --steps 20 --warmup 5
  done
  for f in 14 16 24; do
    run "inflight $f 300" $T --steps 300 --warmup 20 --inflight $f
    run "inflight $f 20" $T --steps 20 --warmup 5 --inflight $f
  done
done
