#!/bin/bash
# Round 4, second GPU visit: suite on the LDS-read-form build, the read-form probe, and an A/B of every bench shape between the
# library built from the previous commit (gpurun_dbg/libgpmpc_hip_base.so, GPMPC_LIB) and the shipped one.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
BASE=$REPO/gpurun_dbg/libgpmpc_hip_base.so
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $OUT/r04b_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04b_parity_report.json 2>/dev/null
tail -3 $OUT/r04b_pytest_gpu_tail.log
timeout 120 tools/microbench/lds_read_forms > $OUT/r04_lds_read_forms.txt 2>&1; cat $OUT/r04_lds_read_forms.txt
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export GPMPC_LIB=$BASE; else unset GPMPC_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline > $OUT/r04b_ab_c2_${v}_$rep.json 2> $OUT/r04b_ab_c2_${v}.err
done
done
for v in base new; do
  if [ $v = base ]; then export GPMPC_LIB=$BASE; else unset GPMPC_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --workload c1 > $OUT/r04b_ab_c1_${v}.json 2> $OUT/r04b_ab_c1_${v}.err
  timeout 300 python bench.py --no-cpu-baseline --workload c3 > $OUT/r04b_ab_c3_${v}.json 2> $OUT/r04b_ab_c3_${v}.err
  timeout 400 python bench.py --no-cpu-baseline --workload c4 > $OUT/r04b_ab_c4_${v}.json 2> $OUT/r04b_ab_c4_${v}.err
  timeout 300 python tools/gpu_c5_late.py 0 2 25 2>&1 | grep "state of" > $OUT/r04b_ab_c5_late_${v}.txt
  timeout 300 python tools/gpu_grad_wide_check.py time 2>&1 | grep -E "WIDE|objective" > $OUT/r04b_ab_c5_grad_${v}.txt
done
unset GPMPC_LIB
python - <<'PY'
import json, glob, os
out = os.environ.get("OUT", "gpurun_out")
for f in sorted(glob.glob("gpurun_out/r04b_ab_c*_*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f grad_ms %s prepare %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], g.get("ms_per_launch"), d["prepare_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
for f in $OUT/r04b_ab_c5_*.txt; do echo $f; cat $f; done
