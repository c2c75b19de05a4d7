#!/bin/bash
# kernel split of gpmpc_prepare at config 5 for the engine options given as name=value arguments
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${TAG:-ab}
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prep_trace_$TAG -o prep -- python $REPO/tools/gpu_prepare_profile.py 4096 16 4 2 "$@" > $OUT/prep_trace_$TAG.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/prep_trace_$TAG/prep_results.db > $OUT/prep_kernel_trace_stats_$TAG.txt
cut -c1-150 $OUT/prep_kernel_trace_stats_$TAG.txt | head -14
python tools/rocpd_summary.py list $OUT/prep_trace_$TAG/prep_results.db t128 > $OUT/prep_dispatches_$TAG.txt
rm -rf $OUT/prep_trace_$TAG
