# tools/big_ab.sh — the big-scene kernels: A/B of one build against the product on config 5's frame, alternating on ONE box
#   usage: tools/big_ab.sh rays1bench_amd/lib/<other>.so
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=rays1bench_amd/lib
OTHER=${1:-$L/librays1_idx64.so}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh.py tests/test_gpu_configs.py -x -q -m gpu -k "big or config5 or 100k or 300k or grid or wavefront" > gpurun_out/big_tests.log 2>&1 || { tail -30 gpurun_out/big_tests.log; exit 1; }
tail -2 gpurun_out/big_tests.log
get='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "%.4f" % d["ms_per_step"])'
run() { local label=$1 so=$2; shift 2; python bench.py --lib $so --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$get" "$label"; }
for i in 1 2 3; do
  for so in $L/librays1.so $OTHER; do
    run "$so config5" $so --scene grid --width 1920 --height 1080 --spp 64 --steps 16 --warmup 4 --inflight 4
  done
done
run "product config5 inflight 8" $L/librays1.so --scene grid --width 1920 --height 1080 --spp 64 --steps 24 --warmup 8 --inflight 8
run "product config5 variant 2" $L/librays1.so --variant 2 --scene grid --width 1920 --height 1080 --spp 4 --steps 4 --warmup 1 --inflight 2
