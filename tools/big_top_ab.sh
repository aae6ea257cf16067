#!/bin/bash
# tuning experiment: nodes of the breadth-first top of the tree the big-scene kernels keep in LDS (R1_BIG_TOP), 100 004-sphere scene
cd $GRAFT_REPO_ROOT
for t in 0 63 127 255 511 0 255; do
  echo -n "R1_BIG_TOP=$t: "; R1_BVH_TOP=1023 R1_BIG_TOP=$t python bench.py --no-cpu-baseline --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s  wg %d' % (d['value'], d['config']['workgroups']))"
done
