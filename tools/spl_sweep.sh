#!/bin/bash
# tools/spl_sweep.sh — samples per lane of the throughput grid (tuning build, R1_SAMPLES_PER_LANE; shipped: 150 -> 254 workgroups per 1200x800x10 frame):
# the long run and the driver's 20 steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
L=rays1bench_amd/lib/librays1_tuning.so
for rep in 1 2; do for spl in 75 100 122 150 200 300; do
  for args in "--steps 20 --warmup 5" "--steps 300 --warmup 20"; do
    echo -n "R1_SAMPLES_PER_LANE=$spl $args: "
    R1_SAMPLES_PER_LANE=$spl timeout -k 10 120 python bench.py --lib $L --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s (%.4f ms, %d workgroups) check %s' % (d['value'], d['ms_per_step'], d['config']['workgroups'], d.get('check')))"
  done
done; done
