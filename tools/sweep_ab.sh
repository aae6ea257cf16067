# tools/sweep_ab.sh — the exhaustive sweep's exact phase: member spheres in group order (R1_EXACT_G) and rays as 16-byte LDS rows
# (R1_RAYS_AOS), each alone and together, alternating on ONE box.  Build first (see the cp lines in the round's notes):
#   librays1.so = both on (product);  librays1_sw00.so / sw10 / sw01 = make tuning EXTRA="-DR1_EXACT_G=a -DR1_RAYS_AOS=b" copied
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or prefilter or reference or big_scene or config5 or sphere_count" > gpurun_out/sweep_tests.log 2>&1 || { tail -30 gpurun_out/sweep_tests.log; exit 1; }
tail -2 gpurun_out/sweep_tests.log
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --variant 2 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2 3; do
  for so in librays1.so librays1_sw00.so librays1_sw10.so librays1_sw01.so; do
    run "$so 300" $L/$so --steps 300 --warmup 20
    run "$so 20" $L/$so --steps 20 --warmup 5
  done
done
