#!/bin/bash
# tools/resolve_rows_ab.sh — rows of workgroups of the resolve launch for frames in flight (tuning build, R1_RESOLVE_ROWS; 0 = one per tile):
# the driver's 20-step command and the long run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
L=rays1bench_amd/lib/librays1_tuning.so
for rep in 1 2; do for rows in 0 16 32 64 128 256; do
  for args in "--steps 20 --warmup 5" "--steps 300 --warmup 20"; do
    echo -n "R1_RESOLVE_ROWS=$rows $args: "
    R1_RESOLVE_ROWS=$rows timeout -k 10 120 python bench.py --lib $L --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s (%.4f ms) check %s' % (d['value'], d['ms_per_step'], d.get('check')))"
  done
done; done
