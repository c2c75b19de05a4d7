#!/bin/bash
# rocprofv3 kernel trace of the gradient path (args: shape B)
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
SHAPE=${1:-c2}; B=${2:-256}
timeout 120 python tools/gpu_grad_profile.py $SHAPE $B 5 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/gpu_grad_profile.py $SHAPE 1 5 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/grad_trace -o grad -- python $REPO/tools/gpu_grad_profile.py $SHAPE $B 5 > $OUT/grad_trace.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/grad_trace/grad_results.db > $OUT/grad_kernel_trace_stats.txt
cut -c1-160 $OUT/grad_kernel_trace_stats.txt | head -14
