# tools/big_knobs.sh — config 5's frame (100 004 spheres, 1920x1080x64) against the big-scene knobs of the tuning build
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { python bench.py --lib $T --no-extras --no-cpu-baseline --scene grid --width 1920 --height 1080 --spp 64 --steps 12 --warmup 4 --inflight 4 2>/dev/null | python -c "$get" "$1"; }
run "base"
for t in 31 95 127; do R1_BIG_TOP=$t R1_BVH_TOP=$t run "top $t"; done
for l in 4 6 12 16; do R1_BVH_LEAF=$l run "leaf $l"; done
for s in 60 100 200; do R1_SAMPLES_PER_LANE=$s run "spl $s"; done
run "base again"
