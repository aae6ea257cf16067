# tools/entry_ab.sh — per-tile entry nodes (R1_ENTRY): parity tests and traversal counters through the build that has them
#   build first:  make -C rays1bench_amd/csrc tuning EXTRA=-DR1_ENTRY=1
#   Grays/s against the product: tools/walk_ab.sh
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
T=rays1bench_amd/lib/librays1_tuning.so
R1_TEST_LIB=$T timeout -k 10 600 python -m pytest tests/test_gpu_bvh.py -x -q -m gpu > gpurun_out/entry_tests.log 2>&1 || { tail -30 gpurun_out/entry_tests.log; exit 1; }
tail -2 gpurun_out/entry_tests.log
R1_LIB=$T R1_ENTRY_PRINT=1 timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/entry_stats_on.txt 2>&1
R1_LIB=$T R1_ENTRY_OFF=1 timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/entry_stats_off.txt 2>&1
head -24 gpurun_out/entry_stats_on.txt
