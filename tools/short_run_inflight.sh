# tools/short_run_inflight.sh — the driver's short command (--steps 20 --warmup 5) against the number of frames in flight
cd $GRAFT_REPO_ROOT
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
for i in 1 2; do
for f in 3 4 5 6 8 10 12 16 20; do
  python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --inflight $f 2>/dev/null | python -c "$get" "inflight $f steps 20"
done
done
for f in 5 6 8 20; do
  python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 --inflight $f 2>/dev/null | python -c "$get" "inflight $f steps 300"
done
