REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/wide_trace -o w -- python $REPO/tools/gpu_grad_wide_check.py time > $OUT/wide_trace.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/wide_trace/w_results.db > $OUT/wide_kernel_trace_stats.txt
cut -c1-150 $OUT/wide_kernel_trace_stats.txt | head -8
rm -rf $OUT/wide_trace
