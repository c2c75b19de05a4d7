#!/bin/bash
# PMC passes on the bench workload (counters only, no tracing)
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq1 -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1
cd $REPO
python tools/rocpd_summary.py pmc $OUT/pmc_sq1/bench_results.db $OUT/pmc_sq2/bench_results.db | grep -E "rollout_kernel|^#|counter" 
