#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning aid: 100 k-sphere scene (BASELINE config 5 shape) vs barrier period and frames in flight
for sc in ${SYNC:-0 4 16 32 128}; do for f in ${INFLIGHT:-4 16}; do
  R1_SYNC_CHUNKS=$sc timeout -k 10 300 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --no-cpu-baseline --scene grid --grid 400x250 --width 480 --height 270 --spp 8 --steps 4 --warmup 1 --inflight $f 2>/dev/null | tail -1 |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sync $sc inflight', d['config']['frames_in_flight'], round(d['value'],2), 'mrays/s ms/step', round(d['ms_per_step'],2), 'valu', round(d['roofline']['valu']['frac'],3))"
done; done
