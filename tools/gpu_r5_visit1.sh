#!/bin/bash
# Round 5, visit 1: the merged build (next/round5 shipped): GPU suite, clock probe, forward A/B of c2 / c3 / c4 against the round-3
# tree (gpurun_dbg/r03tree: the round-3 sources at 498db58 with the library built from them, its own bench.py) interleaved on this
# box, wide-gradient item time, prepare times, per-phase cycles of the config-2 kernel (prof build).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=r05a
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/${T}_build_id.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -rf -x 2>&1 | tail -8 > $OUT/${T}_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/${T}_parity_report.json 2>/dev/null
bash tools/gpu_clock_probe.sh > $OUT/${T}_clock_probe.txt 2>&1
F=$OUT/${T}_forward_ab_vs_r03.txt
: > $F
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4))
except Exception as e: print('$1', 'unreadable', e)"; }
for rep in 1 2; do
  for wl in c3 c4 c2; do
    (cd gpurun_dbg/r03tree && timeout 200 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | line "r03 $wl rep$rep") >> $F
    (timeout 200 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | line "new $wl rep$rep") >> $F
  done
done
cat $F
timeout 600 python tools/gpu_grad_wide_check.py time > $OUT/${T}_wide_grad_time.txt 2>&1
tail -4 $OUT/${T}_wide_grad_time.txt
timeout 300 python tools/gpu_prepare_bench.py 200:3:1 300:3:1 400:4:2 500:2:1 639:3:1 1000:4:2 > $OUT/${T}_prepare_times.txt 2>&1
cat $OUT/${T}_prepare_times.txt
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  for b in 1 256; do
    echo "== fused-horizon kernel, config 2, B = $b: cycles per phase summed over the 25 horizon steps (workgroup 0, prof build)" >> $OUT/${T}_c2_phases.txt
    GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 $b 1 2>&1 | grep -a "PROF cycles\|PROF wave0" | head -3 >> $OUT/${T}_c2_phases.txt
  done
  cat $OUT/${T}_c2_phases.txt
fi
tail -3 $OUT/${T}_pytest_gpu_tail.log
cat $OUT/${T}_clock_probe.txt | tail -5
