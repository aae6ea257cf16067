#!/bin/bash
# tools/sync_sweep.sh — device time of ONE synchronous frame (r1_render) against grid size and
# queue chunk (tuning experiment; env overrides of r1_capi.cpp).  Run on the GPU box.
cd $GRAFT_REPO_ROOT
for b in 1 2 3 4 6; do
  for c in 0 64 1024; do
    echo -n "blocks/CU=$b chunk=$c : "
    R1_BLOCKS_PER_CU=$b R1_CHUNK=$c python tools/variant_times.py large 1200 800 10 --variants=4 2>&1 | grep -o "device [0-9.]* ms"
  done
done
