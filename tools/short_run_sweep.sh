#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning experiment: the driver's command (--steps 20 --warmup 5) and a long run against grid size per frame and frames in flight
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  for rep in 1 2 3; do
    env $1 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'], end=' ')"
  done
  echo -n " | 300 steps: "
  env $1 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --inflight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f  wg %d' % (d['value'], d['config']['workgroups']))"
}
for cfg in ${SWEEP:-"16 150" "16 122" "16 100" "16 75" "16 61" "16 50" "20 122" "20 100" "10 122" "12 100"}; do set -- $cfg
  run "R1_SAMPLES_PER_LANE=$2" $1
done
