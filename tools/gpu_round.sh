#!/bin/bash
# One GPU visit: parity tests, smoke, bench line, rocprofv3 kernel trace + PMC passes.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 3 2>$OUT/bench.err | tee $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/prof_pmc_sq -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $OUT/prof_pmc_sq2 -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_sq2.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/prof_trace/bench_results.db > $OUT/kernel_trace_stats.txt
python tools/rocpd_summary.py pmc $OUT/prof_pmc_fetch/bench_results.db $OUT/prof_pmc_write/bench_results.db $OUT/prof_pmc_sq/bench_results.db $OUT/prof_pmc_sq2/bench_results.db > $OUT/pmc.txt
head -8 $OUT/kernel_trace_stats.txt | cut -c1-140; grep rollout_kernel $OUT/pmc.txt | cut -c1-20,70-140
