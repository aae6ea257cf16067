#!/bin/bash
# tuning experiment: per-rank frame time of an 8-GPU run carried by one GPU (bench.py --emulate-shards) against kernel mode and grid
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env $1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --emulate-shards $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  wg %d' % (d['ms_per_step'], d['config']['workgroups']))"; }
for k in 8 4; do
run "R1_TP_MODE=0" $k
run "R1_TP_MODE=1" $k
run "R1_TP_MODE=1 R1_NQ=1" $k
run "R1_TP_MODE=0 R1_MIN_BLOCKS=128" $k
run "R1_TP_MODE=1 R1_MIN_BLOCKS=128" $k
run "R1_TP_MODE=0 R1_MIN_BLOCKS=96" $k
run "R1_TP_MODE=1 R1_MIN_BLOCKS=96 R1_NQ=4" $k
done
