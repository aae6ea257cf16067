#!/bin/bash
# tools/inproc_sweep.sh — bench.py --multi inproc (one process, r1_multi, one-rank RCCL communicator per frame in flight on a one-GPU box) against frames in flight
for k in 1 8 12 16 20; do timeout -k 10 200 python bench.py --gpus 1 --multi inproc --inflight $k --steps 300 --warmup 20 --check > gpurun_out/b.json 2>gpurun_out/b.err || tail -5 gpurun_out/b.err; python -c "
import json;d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]);print('inproc lanes $k:',round(d['value']),round(d['ms_per_step'],4),d['check'],d['config']['gpu_max_hw_queues'])"; done
