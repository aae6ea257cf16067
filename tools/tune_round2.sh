#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning experiments after the node table moved into LDS and the in-flight walk went back to while-while (round 2):
# spheres per leaf, outlier peeling, grid size per frame (long run and the driver's 20 frames)
R=$GRAFT_REPO_ROOT; cd $R
one() { env $1 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --no-cpu-baseline --steps ${2:-300} --warmup ${3:-20} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"; }
for cfg in "R1_BVH_LEAF=3" "R1_BVH_LEAF=4" "R1_BVH_LEAF=6" "R1_BVH_PEEL=0" "R1_SAMPLES_PER_LANE=100" "R1_SAMPLES_PER_LANE=150" "R1_SAMPLES_PER_LANE=200" "R1_SAMPLES_PER_LANE=300"; do
  echo -n "$cfg : long run "; one $cfg; one $cfg; echo -n " | 20 frames "; one $cfg 20 5; one $cfg 20 5; one $cfg 20 5; echo
done
