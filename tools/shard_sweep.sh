#!/bin/bash
# (the R1_* knobs are only read by the -DR1_TUNING build: make -C rays1bench_amd/csrc tuning)
# tuning experiment: per-rank frame time of a 4- / 8-GPU run carried by one GPU (bench.py --emulate-shards, 16 frames in flight) against queue chunk and grid
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env $1 python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --steps 400 --warmup 20 --no-cpu-baseline --emulate-shards $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  wg %d' % (d['ms_per_step'], d['config']['workgroups']))"; }
for k in 8 4; do
run "R1_TP_MODE=0" $k
run "R1_MIN_BLOCKS=64" $k
run "R1_MIN_BLOCKS=96" $k
run "R1_MIN_BLOCKS=160" $k
run "R1_MIN_BLOCKS=192" $k
run "R1_MIN_BLOCKS=256" $k
run "R1_CHUNK=128" $k
run "R1_CHUNK=512" $k
done
for inflight in 12 20; do echo -n "8 shards, inflight $inflight: "; python bench.py --lib rays1bench_amd/lib/librays1_tuning.so --steps 400 --warmup 20 --no-cpu-baseline --emulate-shards 8 --inflight $inflight 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step' % d['ms_per_step'])"; done
