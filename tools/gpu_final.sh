#!/bin/bash
# Short end-of-round visit: GPU tests, smoke, bench line, rocprofv3 kernel trace + HBM traffic counters (no SQ passes).
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 70 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 60 python bench.py --steps 20 --warmup 3 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 40 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
timeout 30 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_fetch.log 2>&1
timeout 30 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_write.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/prof_trace/bench_results.db > $OUT/kernel_trace_stats.txt
python tools/rocpd_summary.py pmc $OUT/prof_pmc_fetch/bench_results.db $OUT/prof_pmc_write/bench_results.db > $OUT/pmc_traffic.txt
head -4 $OUT/kernel_trace_stats.txt | cut -c1-140; grep rollout_kernel $OUT/pmc_traffic.txt | cut -c1-20,70-140
