#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning experiment: the two pad formulas of the box tree (R1_BVH_PAD_LOCAL) on the large scene and the 100 004-sphere lattice
cd $GRAFT_REPO_ROOT
for m in 0 1; do
  echo -n "large, pad_local=$m: "; R1_BVH_PAD_LOCAL=$m python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['work']; print('%.0f mrays/s  nodes/ray %.2f pairs/ray %.2f' % (d['value'], w['node_visits_per_ray'], w['sphere_pair_tests_per_ray']))"
  echo -n "100k lattice, pad_local=$m: "; R1_BVH_PAD_LOCAL=$m python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --no-cpu-baseline --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['roofline']['work']; print('%.0f mrays/s  nodes/ray %.2f pairs/ray %.2f' % (d['value'], w['node_visits_per_ray'], w['sphere_pair_tests_per_ray']))"
done
