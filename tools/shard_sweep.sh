#!/bin/bash
# tuning experiment: per-rank frame time of an 8-GPU run carried by one GPU (bench.py --emulate-shards) against queue chunk and grid
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env $1 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --emulate-shards $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  wg %d' % (d['ms_per_step'], d['config']['workgroups']))"; }
for k in 8; do
run "R1_TP_MODE=0" $k
run "R1_CHUNK=64" $k
run "R1_CHUNK=128" $k
run "R1_CHUNK=512" $k
run "R1_CHUNK=128 R1_CHUNK_MIN=16" $k
run "R1_MIN_BLOCKS=160" $k
run "R1_MIN_BLOCKS=112" $k
run "R1_MIN_BLOCKS=128 R1_CHUNK=128" $k
done
for inflight in 8 24; do echo -n "inflight $inflight: "; python bench.py --steps 400 --warmup 20 --no-cpu-baseline --emulate-shards 8 --inflight $inflight 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step' % d['ms_per_step'])"; done
