#!/bin/bash
# Round 5, visit 6: reciprocal / inverse root of the per-step small algebra from the hardware seed + Newton steps instead of the IEEE
# division and sqrt-then-divide (P1 of the fused-horizon kernel, serial path of one wavefront); rows-per-chunk sweep on the new build.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=${1:-r05g}
PK=$REPO/data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd
timeout 900 python -m pytest tests -m gpu -q -rf 2>&1 | tail -12 > $OUT/${T}_pytest_gpu_tail.log
tail -4 $OUT/${T}_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/${T}_parity_report.json 2>/dev/null
F=$OUT/${T}_forward_ab.txt
: > $F
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4), '| spread', round(d.get('windows',{}).get('spread',0),4), '| reversed-batch bitwise', ((d.get('parity') or {}).get('batch_independence') or {}).get('bitwise_equal_reversed_batch'), '| dmean', (d.get('parity') or {}).get('max_abs_dmean'), 'cov', (d.get('parity') or {}).get('max_rel_cov'))
except Exception as e: print('$1', 'unreadable', e)"; }
run() {  # lib workload tag extra-args
  L=$REPO/gpurun_dbg/libgpmpc_hip_$1.so
  [ $1 = new ] && L=$PK/libgpmpc_hip.so
  (GPMPC_LIB=$L timeout 200 python bench.py --workload $2 --no-cpu-baseline --no-gradient --steps 20 --warmup 3 $4 2>$OUT/${T}_last.err | line "$1 $2 $3") >> $F
}
for wl in c2 c1 c3 c4; do
  for lib in v4 new v4 new; do run $lib $wl ab; done
done
run new c2 B4096 "--candidates-per-gpu 4096"; run v4 c2 B4096 "--candidates-per-gpu 4096"
cat $F
echo "== fused-horizon kernel, config 2, B = 256: cycles per phase summed over the 25 horizon steps (workgroup 0, prof build)" > $OUT/${T}_c2_phases.txt
GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 256 1 2>&1 | grep -a "PROF cycles\|PROF wave0" | head -3 >> $OUT/${T}_c2_phases.txt
cat $OUT/${T}_c2_phases.txt
tail -3 $OUT/${T}_last.err
