#!/bin/bash
# Round 4, twelfth GPU visit: as the eleventh, after the separable pass split its odd pair over both wavefront pairs (D = 3).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gradient.py tests/test_gpu_controller.py -m gpu -q -rf 2>&1 | tail -12 > $OUT/r04m_pytest_gradient_tail.log
timeout 300 python tools/gpu_grad_sweep.py c2,c3,c2:1024 15 0 > $OUT/r04m_grad_sweep.txt 2> $OUT/r04m_grad_sweep.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04m_g -o g -- python $REPO/tools/gpu_grad_profile.py c2 256 5 > $OUT/r04m_g.log 2>&1
(cd $REPO && python tools/rocpd_summary.py trace $OUT/r04m_g/g_results.db > $OUT/r04m_c2_gradient_kernel_trace_stats.txt 2>&1)
rm -rf $OUT/r04m_g
cd $REPO
tail -3 $OUT/r04m_pytest_gradient_tail.log
cat $OUT/r04m_grad_sweep.txt
head -8 $OUT/r04m_c2_gradient_kernel_trace_stats.txt | cut -c1-130
