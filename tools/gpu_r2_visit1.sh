#!/bin/bash
# Round-2 first GPU visit: parity tests (+ achieved-error report), smoke, FMA-loop probe, bench lines of c2 (default), c1, c3, c4,
# rocprofv3 trace + counters of the shipped c2 kernel.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 60 tools/microbench/mfma_f64_rate > $OUT/fma_loop_microbench.txt 2>&1; grep "8 chains" $OUT/fma_loop_microbench.txt
timeout 300 python bench.py --steps 20 --warmup 3 2>$OUT/bench_c2.err | tee $OUT/bench_c2.json | cut -c1-250
for wl in c1 c3 c4; do
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 2 2>$OUT/bench_$wl.err | tee $OUT/bench_$wl.json | cut -c1-250
done
bash tools/gpu_counters.sh r02_c2 c2:N200:B256 rollout_kernel --workload c2
