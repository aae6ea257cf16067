#!/bin/bash
# tuning experiment: frames with the FULL persistent grid, 1/2/4 in flight, back to back (no host sync between frames)
cd $GRAFT_REPO_ROOT
for f in 1 2 3 4; do
  echo -n "full grid, inflight=$f: "
  R1_MIN_BLOCKS=100000 R1_SAMPLES_PER_LANE=1 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --inflight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['workgroups'], d['roofline']['kernel_ms'])"
done
for wg in 256 512 768; do
  echo -n "wg=$wg inflight=4: "
  R1_MIN_BLOCKS=$wg R1_SAMPLES_PER_LANE=100000 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --inflight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['workgroups'], d['roofline']['kernel_ms'])"
done
