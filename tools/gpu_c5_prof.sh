#!/bin/bash
# Streaming (config-5 class) rollout kernel: parity subset, step times, SQ counters of one N = 2048 launch.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_core or config5 or large_n_variant" 2>&1 | tail -3
timeout 200 python tools/gpu_c5_step.py 2>&1 | grep "N=" | tee $OUT/c5_step_times.txt
cd /tmp && export TMPDIR=/tmp
DBS=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $grp -d $OUT/c5_pmc_$name -o c5 -- python $REPO/tools/gpu_c5_step.py 2048:1 > $OUT/c5_pmc_$name.log 2>&1
  DBS="$DBS $OUT/c5_pmc_$name/c5_results.db"
done
cd $REPO
python tools/rocpd_summary.py pmc $DBS > $OUT/c5_pmc.txt
grep stream $OUT/c5_pmc.txt | cut -c1-30,70-140
# the sqlite outputs are tens of MB each: only the text summaries travel back (gpurun merges at most 64 MiB)
(cd $OUT && rm -rf c5_pmc_SQ_WAVES c5_pmc_SQ_WAIT_INST_LDS)
