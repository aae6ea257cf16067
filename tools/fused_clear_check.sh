#!/bin/bash
# after folding the per-frame memsets into the resolve launch: GPU suite, then the bench line, the driver's command and the emulated per-rank loads
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for a in "--steps 300 --warmup 20" "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 20 --emulate-shards 2" "--steps 200 --warmup 20 --emulate-shards 4" "--steps 200 --warmup 20 --emulate-shards 8"; do
  echo -n "$a : "; python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f mrays/s  %.4f ms/step  submit %.4f' % (d['value'], d['ms_per_step'], d['config']['host_submit_ms_per_step']))"
done
