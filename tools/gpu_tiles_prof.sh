#!/bin/bash
# kernel trace of the batch-major path at config 4 (both paths run once each, B = 2048)
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tiles_trace -o t -- python $REPO/tools/gpu_tiles_check.py c4 "$@" > $OUT/tiles_trace.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/tiles_trace/t_results.db > $OUT/tiles_kernel_trace_stats.txt 2>&1
grep -v "^#" $OUT/tiles_kernel_trace_stats.txt | head -12 | cut -c1-150
tail -3 $OUT/tiles_trace.log
rm -rf $OUT/tiles_trace
