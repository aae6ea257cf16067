#!/bin/bash
# tools/short_batch_sweep.sh — the driver's 20-step command at N = 1: frames per launch x launches in flight (whole frames, r1_render_batch_async)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for rep in 1 2; do for cfg in "1 20" "2 10" "4 5" "5 4" "10 2" "20 1" "7 3" "3 7"; do
  set -- $cfg
  echo -n "batch $1 inflight $2: "
  timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --batch $1 --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%.0f mrays/s  %.4f ms per frame  (%d workgroups per launch) check %s' % (d['value'], d['ms_per_step'], c['workgroups'], d.get('check')))"
done; done
