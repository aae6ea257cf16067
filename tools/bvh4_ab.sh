# tools/bvh4_ab.sh — 4-wide nodes against the binary tree (build first: make -C rays1bench_amd/csrc tuning EXTRA=-DR1_BVH4=1), one box
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
R1_BVH4_PRINT=1 R1_LIB=$T timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/bvh4_stats_wide.txt 2>&1 || { tail -20 gpurun_out/bvh4_stats_wide.txt; exit 1; }
timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/bvh4_stats_binary.txt 2>&1
head -22 gpurun_out/bvh4_stats_wide.txt
R1_TEST_LIB=$T timeout -k 10 600 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/bvh4_tests.log 2>&1 || { tail -30 gpurun_out/bvh4_tests.log; exit 1; }
tail -2 gpurun_out/bvh4_tests.log
for i in 1 2 3; do
for lib in $T rays1bench_amd/lib/librays1.so; do
  echo "== $lib"
  R1_LIB=$lib python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('300:', d['value'], d['ms_per_step'])"
  R1_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20:', d['value'], d['ms_per_step'])"
done
done
