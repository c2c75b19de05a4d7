#!/bin/bash
# Round 4, seventh GPU visit: full suite (incl. the in-range config-5 fixture); config-5 step / wide-gradient item times with the
# split stage fill against the base library; gradient timings of c2 / c3 / c4 (separable moments: version 1 at D <= 3, 2 at D = 4).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
BASE=$REPO/gpurun_dbg/libgpmpc_hip_base.so
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $OUT/r04g_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04g_parity_report.json 2>/dev/null
tail -4 $OUT/r04g_pytest_gpu_tail.log
rm -f $OUT/r04g_ab_*.txt
for v in base new; do
  if [ $v = base ]; then export GPMPC_LIB=$BASE; else unset GPMPC_LIB; fi
  timeout 300 python tools/gpu_c5_late.py 0 2 25 2>&1 | grep "state of" > $OUT/r04g_ab_c5_late_${v}.txt
  timeout 300 python tools/gpu_grad_wide_check.py time 2>&1 | grep -E "WIDE|objective" > $OUT/r04g_ab_c5_grad_${v}.txt
  for wl in "c2 256" "c3 1024" "c4 2048"; do
    set -- $wl
    timeout 300 python tools/gpu_grad_profile.py $1 $2 5 2>&1 | grep "ms per launch" >> $OUT/r04g_ab_grad_${v}.txt
  done
done
unset GPMPC_LIB
for f in $OUT/r04g_ab_*.txt; do echo $f; cat $f; done
