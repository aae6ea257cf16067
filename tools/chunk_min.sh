# tools/chunk_min.sh — smallest chunk a wave takes at the end of a frame's queue (R1_CHUNK_MIN, tuning build)
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
for i in 1 2 3; do
for c in 32 16 64 128; do
  R1_CHUNK_MIN=$c python bench.py --lib $T --no-extras --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "$get" "chunk_min $c steps 300"
  R1_CHUNK_MIN=$c python bench.py --lib $T --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$get" "chunk_min $c steps 20"
done
done
