#!/bin/bash
# tuning experiment: synchronous frame time against queue / tail settings (env overrides of r1_capi.cpp)
cd $GRAFT_REPO_ROOT
V=${1:-4}
run() { echo -n "$* : "; env "$@" python tools/variant_times.py large 1200 800 10 --variants=$V 2>&1 | grep -o "device [0-9.]* ms\|IMAGE DIFFERS"; }
run R1_COOP_LANES=0
run R1_COOP_LANES=4
run R1_COOP_LANES=4 R1_NQ=8
run R1_COOP_LANES=4 R1_NQ=24
run R1_COOP_LANES=4 R1_TILE_ORDER_HACK=3,19
run R1_COOP_LANES=4 R1_TILE_ORDER_HACK=20,25
run R1_COOP_LANES=4 R1_TILE_ORDER_HACK=5,15
