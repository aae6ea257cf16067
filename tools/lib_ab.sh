#!/bin/bash
# tools/lib_ab.sh TAG... — throughput / synchronous rates of several builds of librays1.so (rays1bench_amd/lib/librays1_TAG.so.bak)
R=$GRAFT_REPO_ROOT; cd $R
cp rays1bench_amd/lib/librays1.so /tmp/keep.so
for rep in 1 2; do for tag in "$@"; do
  cp rays1bench_amd/lib/librays1_$tag.so.bak rays1bench_amd/lib/librays1.so
  echo -n "$tag: "
  python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight %.0f mrays/s (%.4f ms)  sync device %.4f ms  sweep %.0f' % (d['value'], d['ms_per_step'], d['value_dispatch_to_host']['device_ms_per_step'], d['exhaustive_sweep']['value']))"
done; done
cp /tmp/keep.so rays1bench_amd/lib/librays1.so
