#!/bin/bash
# tuning experiment after the node table moved into LDS: carry fraction (builds cd3/cd4/cd6), spheres per leaf, then the GPU suite
R=$GRAFT_REPO_ROOT; cd $R
bash tools/lib_ab.sh cd4 cd3 cd6
for leaf in 3 4 6 8; do echo -n "leaf $leaf: "; R1_BVH_LEAF=$leaf python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s nodes %d depth %d' % (d['value'], d['config']['bvh']['nodes'], d['config']['bvh']['depth']))"; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
