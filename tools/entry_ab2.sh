# tools/entry_ab2.sh — (build first: make -C rays1bench_amd/csrc tuning EXTRA=-DR1_ENTRY=1) per-tile entry nodes on / off on the other BASELINE configurations (librays1_tuning.so, knob R1_ENTRY_OFF)
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export R1_LIB=rays1bench_amd/lib/librays1_tuning.so
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["ms_per_step"], "sync", d.get("value_synchronous_frame", {}).get("value") if isinstance(d.get("value_synchronous_frame"), dict) else d.get("value_synchronous_frame"))'
for i in 1 2; do
for off in 0 1; do
  echo "== R1_ENTRY_OFF=$off"
  R1_ENTRY_PRINT=1 R1_ENTRY_OFF=$off timeout -k 10 200 python bench.py --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4 --no-cpu-baseline --no-extras 2>gpurun_out/e.err | python -c "$get" config5; grep "entry nodes" gpurun_out/e.err | head -1 || true
  R1_ENTRY_OFF=$off timeout -k 10 200 python bench.py --scene medium --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$get" medium
  R1_ENTRY_OFF=$off timeout -k 10 200 python bench.py --spp 250 --steps 48 --warmup 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$get" spp250
  R1_ENTRY_OFF=$off timeout -k 10 200 python bench.py --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$get" inflight1
done
done
