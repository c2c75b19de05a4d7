#!/bin/bash
# Round 4, sixth GPU visit: separable gradient moments, version 2 with the chunk inputs loaded one chunk ahead -- gradient tests, then
# gradient timings of c2 / c3 / c4 against the library of the commit before version 2 (gpurun_dbg/libgpmpc_hip_base.so).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
BASE=$REPO/gpurun_dbg/libgpmpc_hip_base.so
timeout 900 python -m pytest tests/test_gpu_gradient.py tests/test_gpu_controller.py -m gpu -q -rf 2>&1 | tail -30 > $OUT/r04f_pytest_gradient_tail.log
tail -3 $OUT/r04f_pytest_gradient_tail.log
for v in base new; do
  if [ $v = base ]; then export GPMPC_LIB=$BASE; else unset GPMPC_LIB; fi
  for wl in "c2 256" "c3 1024" "c4 2048"; do
    set -- $wl
    timeout 300 python tools/gpu_grad_profile.py $1 $2 5 2>&1 | grep "ms per launch" >> $OUT/r04f_ab_grad_${v}.txt
  done
done
unset GPMPC_LIB
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04f_gc4 -o g -- python $REPO/tools/gpu_grad_profile.py c4 2048 > $OUT/r04f_gc4.log 2>&1
(cd $REPO && python tools/rocpd_summary.py trace $OUT/r04f_gc4/g_results.db > $OUT/r04f_c4_gradient_kernel_trace_stats.txt 2>&1)
rm -rf $OUT/r04f_gc4
cd $REPO
for v in base new; do echo == $v; cat $OUT/r04f_ab_grad_${v}.txt; done
head -7 $OUT/r04f_c4_gradient_kernel_trace_stats.txt | cut -c1-150
