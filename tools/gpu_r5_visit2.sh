#!/bin/bash
# Round 5, visit 2: the compact work-item list of the fused-horizon kernel (no empty queue slots) and the paired per-point pass,
# A/B against the library before them on one box (GPMPC_LIB): base = HEAD e8f0a47, list = item list only, new = list + paired pass.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=r05b
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/${T}_build_id.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -rf -x 2>&1 | tail -8 > $OUT/${T}_pytest_gpu_tail.log
tail -3 $OUT/${T}_pytest_gpu_tail.log
F=$OUT/${T}_forward_ab.txt
: > $F
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', '| rollouts/s', round(d['value']), '| ms/step', round(d['ms_per_step'],4), '| kernel ms', round(d['roofline']['kernel_ms'],4), '| windows', [round(x,4) for x in d.get('windows',{}).get('ms_per_step',[])])
except Exception as e: print('$1', 'unreadable', e)"; }
for rep in 1 2; do
  for wl in c2 c1 c3 c4; do
    for lib in base list new; do
      L=$REPO/gpurun_dbg/libgpmpc_hip_$lib.so
      [ $lib = new ] && L=$REPO/data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd/libgpmpc_hip.so
      (GPMPC_LIB=$L timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-gradient --steps 20 --warmup 3 2>/dev/null | line "$lib $wl rep$rep") >> $F
    done
  done
done
cat $F
F=$OUT/${T}_rows_per_chunk.txt
: > $F
for ch in 24 28 32 36 40 48 64; do
  (timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-gradient --steps 20 --warmup 3 --option rows_per_chunk=$ch 2>/dev/null | line "new c2 rows_per_chunk=$ch") >> $F
done
cat $F
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  echo "== fused-horizon kernel, config 2, B = 256: cycles per phase summed over the 25 horizon steps (workgroup 0, prof build of the new sources)" > $OUT/${T}_c2_phases.txt
  GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 256 1 2>&1 | grep -a "PROF cycles\|PROF wave0" | head -3 >> $OUT/${T}_c2_phases.txt
  cat $OUT/${T}_c2_phases.txt
fi
