#!/bin/bash
# batch-major path: parity, timings, kernel trace at config 4
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python tools/gpu_tiles_check.py parity time "$@" > $OUT/tiles_check.log 2>&1
grep -E "FAIL|PARITY|bitwise|rollouts/s|tile_chunk|Error|error" $OUT/tiles_check.log | head -40
bash tools/gpu_tiles_prof.sh "$@"
