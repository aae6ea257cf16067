#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning experiments on the 100 004-sphere scene (big-scene tree kernel, 8 waves per SIMD): nodes of the tree's top kept in LDS
# (R1_BIG_TOP), spheres per leaf (R1_BVH_LEAF)
cd $GRAFT_REPO_ROOT
one() { env $* python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --no-cpu-baseline --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s  wg %d  depth %d' % (d['value'], d['config']['workgroups'], d['config']['bvh']['depth']))"; }
for cfg in "R1_BIG_TOP=0" "R1_BIG_TOP=31" "R1_BIG_TOP=63" "R1_BIG_TOP=127 R1_BVH_TOP=127" "R1_BVH_LEAF=6" "R1_BVH_LEAF=8" "R1_BVH_LEAF=10" "R1_BVH_LEAF=12"; do echo -n "$cfg: "; one $cfg; done
