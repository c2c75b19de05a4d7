#!/bin/bash
# Round 5, visit 3: fused-horizon kernel -- compact work-item list (lane-parallel builder), paired per-point pass, tabulated lane
# assignment of the diagonal pairs' items.  A/B on one box (GPMPC_LIB): base = e8f0a47, list = first list version (serial builder,
# serial per-point pass), new = everything, nolm = new without the lane table, serp2 = new with the serial per-point pass.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=r05c
PK=$REPO/data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/${T}_build_id.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -rf -x 2>&1 | tail -8 > $OUT/${T}_pytest_gpu_tail.log
tail -3 $OUT/${T}_pytest_gpu_tail.log
F=$OUT/${T}_forward_ab.txt
: > $F
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4), '| windows', [round(x,4) for x in d.get('windows',{}).get('ms_per_step',[])], '| batch-indep', (d.get('parity') or {}).get('batch_independence'))
except Exception as e: print('$1', 'unreadable', e)"; }
run() {  # lib workload tag extra-args
  L=$REPO/gpurun_dbg/libgpmpc_hip_$1.so
  [ $1 = new ] && L=$PK/libgpmpc_hip.so
  (GPMPC_LIB=$L timeout 200 python bench.py --workload $2 --no-cpu-baseline --no-gradient --steps 20 --warmup 3 $4 2>$OUT/${T}_last.err | line "$1 $2 $3") >> $F
}
for rep in 1 2; do
  for lib in base list new nolm serp2; do run $lib c2 rep$rep; done
done
for wl in c1 c3 c4; do
  for lib in base new base new; do run $lib $wl ab; done
done
cat $F
F2=$OUT/${T}_rows_per_chunk.txt
F=$F2
: > $F
for ch in 24 28 32 36 40 48 64; do run new c2 rows_per_chunk=$ch "--option rows_per_chunk=$ch"; done
cat $F
for v in prof prof_serialp2; do
  echo "== fused-horizon kernel, config 2, B = 256: cycles per phase summed over the 25 horizon steps (workgroup 0, build $v)" >> $OUT/${T}_c2_phases.txt
  GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_$v.so timeout 120 python tools/gpu_grad_profile.py c2 256 1 2>&1 | grep -a "PROF cycles\|PROF wave0" | head -3 >> $OUT/${T}_c2_phases.txt
done
cat $OUT/${T}_c2_phases.txt
tail -5 $OUT/${T}_last.err
