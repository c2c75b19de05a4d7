#!/bin/bash
# Round 4, fifth GPU visit: suite on the build with the second version of the separable gradient moments (precomputed weightings,
# pair-product monomial tables, 32-point chunks); gradient timings of every shape; kernel traces of the c2 / c4 gradient.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $OUT/r04e_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04e_parity_report.json 2>/dev/null
tail -4 $OUT/r04e_pytest_gpu_tail.log
for wl in c1 c2 c3 c4; do
  timeout 400 python bench.py --no-cpu-baseline --workload $wl > $OUT/r04e_$wl.json 2> $OUT/r04e_$wl.err
done
cd /tmp && export TMPDIR=/tmp
for wl in "c2 256" "c4 2048"; do
  set -- $wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04e_g$1 -o g -- python $REPO/tools/gpu_grad_profile.py $1 $2 > $OUT/r04e_g$1.log 2>&1
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/r04e_g$1/g_results.db > $OUT/r04e_$1_gradient_kernel_trace_stats.txt 2>&1)
  rm -rf $OUT/r04e_g$1
done
cd $REPO
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04e_c?.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f grad_ms %s prepare %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], g.get("ms_per_launch"), d["prepare_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
for wl in c2 c4; do head -9 $OUT/r04e_${wl}_gradient_kernel_trace_stats.txt | cut -c1-150; done
