#!/bin/bash
# tools/lib_ab.sh TAG... — throughput / synchronous rates of several builds of the library, rays1bench_amd/lib/librays1_TAG.so
# (built e.g. with `make -C rays1bench_amd/csrc tuning EXTRA=-DR1_...` and copied to a tag).  Every build is loaded
# explicitly (bench.py --lib): the shipped librays1.so is never touched.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for rep in 1 2; do for tag in "$@"; do
  echo -n "$tag: "
  python bench.py --lib rays1bench_amd/lib/librays1_$tag.so --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('to host %.0f mrays/s (%.4f ms)  resident %.0f  sync device %.4f ms  sweep %.0f' % (d['value'], d['ms_per_step'], d['value_device_resident']['value'], d['value_dispatch_to_host']['device_ms_per_step'], d['exhaustive_sweep']['value']))"
done; done
