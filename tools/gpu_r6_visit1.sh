#!/bin/bash
# Round 6, visit 1: the few-candidate cooperative form -- bitwise check + times, B = 1 kernel trace, phase cycles (prof build).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=r06a
timeout 900 python tools/gpu_cluster_check.py > $OUT/${T}_cluster_check.txt 2>&1
tail -40 $OUT/${T}_cluster_check.txt
cd /tmp && export TMPDIR=/tmp
F=$OUT/${T}_b1_latency.txt
: > $F
for cs in 1 0; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/b1_$cs -o g -- python $REPO/tools/gpu_grad_profile.py c2 1 20 cluster=$cs > $OUT/b1_$cs.log 2>&1
  echo "== config 2, B = 1, option cluster = $cs: wall clock per launch (20 launches), then the kernel trace" >> $F
  grep -a "ms per launch" $OUT/b1_$cs.log >> $F
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/b1_$cs/g_results.db | head -9 | cut -c1-150) >> $F
  rm -rf $OUT/b1_$cs
done
cd $REPO
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  for cs in 1 0; do
    echo "== prof build, B = 1, option cluster = $cs: cycles per phase summed over the horizon (workgroup 0)" >> $F
    GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 1 1 cluster=$cs 2>&1 | grep -a "PROF" | head -12 >> $F
  done
fi
cat $F
