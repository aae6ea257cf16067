#!/bin/bash
# tools/ab_lds.sh — experiment: node table in LDS / attenuation stack split between LDS and the global workspace (builds tagged f8, g17, g8)
R=$GRAFT_REPO_ROOT; cd $R
cp rays1bench_amd/lib/librays1.so /tmp/keep.so
for rep in 1 2; do for cfg in "f4 1" "f6 1" "f8 1" "f12 1"; do set -- $cfg
  cp rays1bench_amd/lib/librays1_$1.so.bak rays1bench_amd/lib/librays1.so
  echo -n "$1 nodes_lds=$2: "
  R1_NODES_LDS=$2 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight %.0f mrays/s (%.4f ms)  sync device %.4f ms  rays %d' % (d['value'], d['ms_per_step'], d['value_dispatch_to_host']['device_ms_per_step'], d['config']['rays_per_step']))"
done; done
cp /tmp/keep.so rays1bench_amd/lib/librays1.so
