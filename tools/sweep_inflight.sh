#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning aid: per-rank frame time of a K-GPU run (shard 0 of K on one GPU) vs frames in flight
# and samples per lane (grid size)
for k in ${SHARDS:-8}; do for spl in ${SPL:-9 18 36 72}; do for f in ${INFLIGHT:-4 8 16}; do
  R1_SAMPLES_PER_LANE=$spl timeout -k 10 200 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --steps 200 --warmup 20 --no-cpu-baseline --inflight $f --emulate-shards $k 2>/dev/null | tail -1 |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('shards $k spl $spl inflight', d['config']['frames_in_flight'], 'blocks', d['config']['workgroups'], 'ms/step', round(d['ms_per_step'],3))"
done; done; done
