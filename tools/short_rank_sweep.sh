#!/bin/bash
# tools/short_rank_sweep.sh — the driver's 20-step command through the rank path on ONE GPU carrying rank 0's share of a K-GPU run
# (--emulate-shards K --rccl-selftest): frames per launch x launches in flight; ms per frame (ideal = the one-GPU frame time / K)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for K in 8 4 2; do
  for cfg in "0 0" "1 10" "1 12" "2 5" "2 10" "3 7" "4 5" "5 4" "10 2" "20 1"; do
    set -- $cfg
    echo -n "shards $K batch $1 inflight $2: "
    timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --emulate-shards $K --rccl-selftest --batch $1 --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%.4f ms per frame  (%d workgroups, %d frames per launch, %d launches in flight)' % (d['ms_per_step'], c['workgroups'], c['frames_per_launch'], c['launches_in_flight']))"
  done
done
