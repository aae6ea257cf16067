#!/bin/bash
# tuning experiment: the driver's command (--steps 20 --warmup 5) against grid size per frame and frames in flight
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  for rep in 1 2 3; do
    env $1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"
  done
  echo -n " | 300 steps: "
  env $1 python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f  wg %d' % (d['value'], d['config']['workgroups']))"
}
for inflight in 2 3 4; do for mb in 512 768 1024 1536; do
  run "R1_MIN_BLOCKS=$mb R1_SAMPLES_PER_LANE=100000 R1_LAT_BLOCKS=400 GPU_MAX_HW_QUEUES=$inflight" $inflight
done; done
run "R1_MIN_BLOCKS=768 R1_SAMPLES_PER_LANE=100000 GPU_MAX_HW_QUEUES=3" 3
run "R1_MIN_BLOCKS=256 R1_SAMPLES_PER_LANE=100000" 16
