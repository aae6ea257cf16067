#!/bin/bash
# tools/lib_ab5.sh TAG... — like lib_ab.sh for the 100 004-sphere scene (big-scene tree kernel) and the large scene at 250 spp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for tag in "$@"; do
  echo -n "$tag: 100k lattice "
  python bench.py --lib rays1bench_amd/lib/librays1_$tag.so --no-cpu-baseline --no-extras --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s' % d['value'], end='  ')"
  echo -n "| large x 250 spp "
  python bench.py --lib rays1bench_amd/lib/librays1_$tag.so --no-cpu-baseline --no-extras --spp 250 --steps 48 --warmup 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s' % d['value'])"
done
