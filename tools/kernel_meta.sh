#!/bin/bash
# tools/kernel_meta.sh — register / LDS / spill metadata + flat_load / scratch / MFMA census of every kernel in librays1.so (see kernel_meta.py)
exec python3 "$(dirname "$0")/kernel_meta.py" "$@"
