#!/bin/bash
# One candidate at a time (the reference's restarts_optim 1-2 regime): kernel trace of gpmpc_rollout / gpmpc_rollout_grad at B = 1 and
# B = 2, and the per-phase cycle counters of the fused-horizon kernel (prof build) -- the serial share of a horizon step bounds what
# spreading one candidate's pairwise work over several CUs could gain.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F=$OUT/r04z_b1_latency.txt
: > $F
for b in 1 2; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/b1_$b -o g -- python $REPO/tools/gpu_grad_profile.py c2 $b 20 > $OUT/b1_$b.log 2>&1
  echo "== config 2, B = $b: wall clock per launch (20 launches), then the kernel trace" >> $F
  grep -a "ms per launch" $OUT/b1_$b.log >> $F
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/b1_$b/g_results.db | head -9 | cut -c1-140) >> $F
  rm -rf $OUT/b1_$b
done
cd $REPO
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  echo "== fused-horizon kernel, B = 1: cycles per phase summed over the 25 horizon steps (workgroup 0, prof build)" >> $F
  GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 1 1 grad_mean=1 2>&1 | grep -a "PROF cycles\|PROF wave0\|PROF moments" | head -6 >> $F
fi
cat $F
