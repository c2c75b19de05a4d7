#!/bin/bash
# phase cycles of the fused-horizon kernel (prof build) at B = 1: plain vs cooperative form; args: shape [cluster sizes...]
REPO=$PWD
SHAPE=${1:-c2}; shift
for cs in ${@:-1 0}; do
  echo "== prof build, $SHAPE B = 1, option cluster = $cs"
  GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py $SHAPE 1 1 cluster=$cs 2>&1 | grep -a "PROF\|ms per launch" | sort | uniq | head -60
done
