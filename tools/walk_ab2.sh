# tools/walk_ab2.sh — per-tile entry nodes with the table in global memory (the LDS copy costs the seventh workgroup per CU), against the product
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { label=$1; lib=$2; shift 2; python bench.py --lib $lib --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
export R1_ENTRY_LDS=0
for i in 1 2 3; do
  run "product 300" $L/librays1.so --steps 300 --warmup 20
  run "entry(global) 300" $L/librays1_entry.so --steps 300 --warmup 20
  R1_ENTRY_OFF=1 run "entry(global, table off) 300" $L/librays1_entry.so --steps 300 --warmup 20
  run "product 20" $L/librays1.so --steps 20 --warmup 5
  run "entry(global) 20" $L/librays1_entry.so --steps 20 --warmup 5
  R1_ENTRY_OFF=1 run "entry(global, table off) 20" $L/librays1_entry.so --steps 20 --warmup 5
done
for lib in librays1.so librays1_entry.so; do
  run "$lib medium" $L/$lib --scene medium --steps 300 --warmup 20
  run "$lib spp250" $L/$lib --spp 250 --steps 48 --warmup 16
done
