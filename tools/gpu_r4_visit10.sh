#!/bin/bash
# Round 4, tenth GPU visit (as the ninth, after the set-up phases moved: mean part by mean_moments_kernel, pair solves beside the loads): the gradient's LDS-resident moment pass after the schedule-model / fold / two-workgroups-per-CU work --
# whole GPU suite, the option sweep (tools/gpu_grad_sweep.py), kernel traces of the config-2 gradient with and without the shared CU,
# and the per-phase cycle counters of the moment kernel (prof build).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/r04k_build_id.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -25 > $OUT/r04k_pytest_gpu_tail.log
timeout 600 python tools/gpu_grad_sweep.py c2,c1:2048,c3 15 0,64,48,40,32 > $OUT/r04k_grad_sweep.txt 2> $OUT/r04k_grad_sweep.err
timeout 300 python tools/gpu_grad_sweep.py c2:1024,c2:4096 10 0,64 >> $OUT/r04k_grad_sweep.txt 2>> $OUT/r04k_grad_sweep.err
cd /tmp && export TMPDIR=/tmp
for v in "auto" "share grad_share_cu=1"; do
  set -- $v
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04k_g_$tag -o g -- python $REPO/tools/gpu_grad_profile.py c2 256 5 "$@" > $OUT/r04k_g_$tag.log 2>&1
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/r04k_g_$tag/g_results.db > $OUT/r04k_c2_gradient_${tag}_kernel_trace_stats.txt 2>&1)
  rm -rf $OUT/r04k_g_$tag
done
cd $REPO
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  for v in "" "grad_share_cu=1" "grad_chunk_rows=64"; do
    GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 256 1 $v 2>&1 | grep -a "PROF moments\|ms per launch" | head -6 | sed "s/^/[$v] /" >> $OUT/r04k_moment_phases.txt
  done
fi
tail -3 $OUT/r04k_pytest_gpu_tail.log
cat $OUT/r04k_grad_sweep.txt
cat $OUT/r04k_moment_phases.txt 2>/dev/null
for t in auto share; do head -8 $OUT/r04k_c2_gradient_${t}_kernel_trace_stats.txt | cut -c1-130; done
