#!/bin/bash
# tools/pmc_variant.sh VARIANT SCENEARGS... — PMC counters of one variant's trace kernel, one
# rocprofv3 pass per counter group (measurement tool; run on the GPU box).
# usage: tools/pmc_variant.sh 4 large 1200 800 10        (PASSES="sq1 sq2 ta1 ta2 tc1 tc2" by default)
set -e
cd /tmp && export TMPDIR=/tmp
V=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_v$V
rm -rf $OUT && mkdir -p $OUT
declare -A G
G[sq1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS"
# the texture-addresser / cache blocks have two counter slots each: two counters per pass
G[ta1]="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
G[ta2]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
G[tc1]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
G[tc2]="TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
for tag in ${PASSES:-sq1 sq2 ta1 ta2 tc1 tc2}; do
  timeout -k 5 90 rocprofv3 --pmc ${G[$tag]} --kernel-trace -d $OUT/$tag -o $tag --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/variant_times.py "$@" --variants=$V > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "r1_trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-40s %16.0f  (mean of %d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
