# tools/bvh4_ab.sh — 4-wide nodes (R1_BVH4): traversal counters and every GPU parity test through the build that has them
#   build first:  make -C rays1bench_amd/csrc tuning EXTRA=-DR1_BVH4=1
#   Grays/s against the product: tools/walk_ab.sh
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
R1_BVH4_PRINT=1 R1_LIB=$T timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/bvh4_stats_wide.txt 2>&1 || { tail -20 gpurun_out/bvh4_stats_wide.txt; exit 1; }
timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/bvh4_stats_binary.txt 2>&1
head -22 gpurun_out/bvh4_stats_wide.txt
R1_TEST_LIB=$T timeout -k 10 600 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/bvh4_tests.log 2>&1 || { tail -30 gpurun_out/bvh4_tests.log; exit 1; }
tail -2 gpurun_out/bvh4_tests.log
