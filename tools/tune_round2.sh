#!/bin/bash
# tuning experiments after the node table moved into LDS (round 2): latency-mode knobs of one synchronous frame, then
# grid size per frame x frames in flight for the driver's short run and a long run
R=$GRAFT_REPO_ROOT; cd $R
python tools/ab_sync.py "R1_COOP_LANES=4" "R1_COOP_LANES=2" "R1_COOP_LANES=8" "R1_COOP_LANES=16" "R1_NQ=8" "R1_NQ=32" "R1_CHUNK=128" "R1_CHUNK=32" "R1_BLOCKS_PER_CU=5" --reps=3
SWEEP='"16 150" "20 150" "24 150" "20 122" "24 122" "20 100" "24 100"' 
for cfg in "16 150" "20 150" "24 150" "20 122" "24 122" "20 100" "24 100"; do set -- $cfg
  echo -n "inflight $1 samples/lane $2 : "
  for rep in 1 2 3; do
    R1_SAMPLES_PER_LANE=$2 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --inflight $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"
  done
  echo -n " | 300 steps: "
  R1_SAMPLES_PER_LANE=$2 python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --inflight $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f  wg %d' % (d['value'], d['config']['workgroups']))"
done
