#!/bin/bash
# Round 4, third GPU visit: suite on the build with the fused tile moments (config-4 gradient) and the odd stage stride (config 5);
# A/B: config-4 bench with / without the fusion (same library, option grad_fuse), config-5 step times and wide-gradient item times
# against the library of the round's first commit (gpurun_dbg/libgpmpc_hip_base.so).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
BASE=$REPO/gpurun_dbg/libgpmpc_hip_base.so
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $OUT/r04c_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04c_parity_report.json 2>/dev/null
tail -4 $OUT/r04c_pytest_gpu_tail.log
timeout 400 python bench.py --no-cpu-baseline --workload c4 > $OUT/r04c_c4_fused.json 2> $OUT/r04c_c4_fused.err
timeout 400 python bench.py --no-cpu-baseline --workload c4 --option grad_fuse=0 > $OUT/r04c_c4_separate.json 2> $OUT/r04c_c4_separate.err
timeout 300 python bench.py --no-cpu-baseline --workload c3 > $OUT/r04c_c3.json 2> $OUT/r04c_c3.err
for v in new base; do
  if [ $v = base ]; then export GPMPC_LIB=$BASE; else unset GPMPC_LIB; fi
  timeout 300 python tools/gpu_c5_late.py 0 2 25 2>&1 | grep "state of" > $OUT/r04c_ab_c5_late_${v}.txt
  timeout 300 python tools/gpu_grad_wide_check.py time 2>&1 | grep -E "WIDE|objective" > $OUT/r04c_ab_c5_grad_${v}.txt
done
unset GPMPC_LIB
# kernel trace of the fused config-4 gradient
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04c_c4g_trace -o g -- python $REPO/tools/gpu_grad_profile.py c4 2048 > $OUT/r04c_c4g_trace.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/r04c_c4g_trace/g_results.db > $OUT/r04c_c4_gradient_kernel_trace_stats.txt 2>&1
rm -rf $OUT/r04c_c4g_trace
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04c_c*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f grad_ms %s prepare %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], g.get("ms_per_launch"), d["prepare_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
for f in $OUT/r04c_ab_c5_*.txt; do echo $f; cat $f; done
head -12 $OUT/r04c_c4_gradient_kernel_trace_stats.txt | cut -c1-160
