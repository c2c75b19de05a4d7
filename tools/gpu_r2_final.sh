#!/bin/bash
# Round-2 evidence run: parity suite (+ achieved-error report), smoke, FMA / MFMA probe, bench lines of every BASELINE shape,
# rocprofv3 kernel trace + counters of the config-2 kernel, counters of the streaming kernel, prepare trace at config 5.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $OUT/pytest_gpu_tail.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 60 tools/microbench/mfma_f64_rate > $OUT/fma_loop_microbench.txt 2>&1
bash tools/gpu_counters.sh r02_c2 c2:N200:B256 rollout_kernel --workload c2 > $OUT/counters_c2.log 2>&1
cp $OUT/pmc_counters.json $OUT/pmc_traffic.json profiles/ 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 2>$OUT/bench_c2.err | tee $OUT/bench_c2.json | cut -c1-200
for wl in c1 c3 c4; do
  timeout 400 python bench.py --workload $wl --steps 10 --warmup 2 2>$OUT/bench_$wl.err | tee $OUT/bench_$wl.json | cut -c1-200
done
timeout 900 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 --cpu-seconds 6 2>$OUT/bench_c5.err | tee $OUT/bench_c5.json | cut -c1-200
timeout 200 python tools/gpu_prepare_bench.py 50:3:1 200:3:1 500:2:1 1000:4:2 4096:16:4 2>&1 | grep prepare | tee $OUT/prepare_times.txt
bash tools/gpu_prepare_prof.sh > /dev/null 2>&1
bash tools/gpu_c5_prof.sh > $OUT/c5_prof.log 2>&1
