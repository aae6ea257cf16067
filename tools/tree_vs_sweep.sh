#!/bin/bash
# tuning experiment: box tree against the exhaustive sweep on the small / medium scenes and on sphere-count slices of the lattice
cd $GRAFT_REPO_ROOT
for scene in small medium; do for v in 0 2 4; do
  echo -n "$scene variant $v: "; python bench.py --no-cpu-baseline --scene $scene --variant $v --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s  kernel %s' % (d['value'], d['config']['kernel']))"
done; done
