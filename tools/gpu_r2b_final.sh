#!/bin/bash
# Round-2 (second half) evidence run on the shipped build: parity suite (+ achieved-error report), smoke, FMA / MFMA probe,
# bench lines of every BASELINE shape, config-5 step times, counters of the streaming kernel.  (Counters of the c1-c4 rollout
# kernels and the config-5 prepare trace: tools/gpu_counters.sh, tools/gpu_prepare_prof.sh, run separately.)
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest_gpu_tail.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 60 tools/microbench/mfma_f64_rate > $OUT/fma_loop_microbench.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 2>$OUT/bench_c2.err | tee $OUT/bench_c2.json | cut -c1-200
for wl in c1 c3 c4; do
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 2 2>$OUT/bench_$wl.err | tee $OUT/bench_$wl.json | cut -c1-200
done
timeout 900 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 --cpu-seconds 6 2>$OUT/bench_c5.err | tee $OUT/bench_c5.json | cut -c1-200
bash tools/gpu_c5_prof.sh > $OUT/c5_prof.log 2>&1
timeout 300 python tools/gpu_c5_step.py 4096:1 force_path=4 2>&1 | grep "N=" | sed 's/$/  (force_path=4: tabulated exp)/' >> $OUT/c5_step_times.txt
