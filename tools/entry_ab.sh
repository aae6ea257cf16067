# tools/entry_ab.sh — (build first: make -C rays1bench_amd/csrc tuning EXTRA=-DR1_ENTRY=1) per-tile entry nodes on / off (and the table in LDS / in global memory) in ONE build (librays1_tuning.so, knobs
# R1_ENTRY_OFF, R1_ENTRY_LDS), alternating on one box
set -e; mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R1_TEST_LIB=rays1bench_amd/lib/librays1_tuning.so timeout -k 10 600 python -m pytest tests/test_gpu_bvh.py -x -q -m gpu > gpurun_out/entry_tests.log 2>&1 || { tail -30 gpurun_out/entry_tests.log; exit 1; }
tail -2 gpurun_out/entry_tests.log
export R1_LIB=rays1bench_amd/lib/librays1_tuning.so
R1_ENTRY_PRINT=1 timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/entry_stats_on.txt 2>&1
R1_ENTRY_OFF=1 timeout -k 10 120 python tools/bvh_stats.py large 1200 800 10 > gpurun_out/entry_stats_off.txt 2>&1
for i in 1 2 3; do
for cfg in "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== R1_ENTRY_OFF=$1 R1_ENTRY_LDS=$2"
  R1_ENTRY_OFF=$1 R1_ENTRY_LDS=$2 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('300:', d['value'], d['ms_per_step'])"
  R1_ENTRY_OFF=$1 R1_ENTRY_LDS=$2 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20:', d['value'], d['ms_per_step'])"
done
done
