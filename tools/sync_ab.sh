# tools/sync_ab.sh — one synchronous frame (r1_render span): a build against the product, alternating on ONE box
#   usage: tools/sync_ab.sh rays1bench_amd/lib/<other>.so
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
OTHER=${1:-$L/librays1_lat8.so}
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["value_dispatch_to_host"]; print(sys.argv[1], round(s["value"]), "wall %.4f device %.4f" % (s["ms_per_step"], s["device_ms_per_step"]))'
for i in 1 2 3; do
  for so in $L/librays1.so $OTHER; do
    python bench.py --lib $so --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$get" "$so"
  done
done
