#!/bin/bash
# gpmpc_prepare at config 5 (N = 4096, D = 16): kernel split (rocprofv3 --kernel-trace --stats) and, in separate passes,
# the SQ counters of the matrix-core kernels.  Text summaries only travel back (the sqlite outputs are tens of MB).
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prep_trace -o prep -- python $REPO/tools/gpu_prepare_profile.py 4096 16 4 2 > $OUT/prep_trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/prep_pmc_mfma -o prep -- python $REPO/tools/gpu_prepare_profile.py 4096 16 4 1 > $OUT/prep_pmc_mfma.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY -d $OUT/prep_pmc_lds -o prep -- python $REPO/tools/gpu_prepare_profile.py 4096 16 4 1 > $OUT/prep_pmc_lds.log 2>&1
cd $REPO
python tools/rocpd_summary.py trace $OUT/prep_trace/prep_results.db > $OUT/prep_kernel_trace_stats.txt
python tools/rocpd_summary.py pmc $OUT/prep_pmc_mfma/prep_results.db $OUT/prep_pmc_lds/prep_results.db > $OUT/prep_pmc.txt
cut -c1-150 $OUT/prep_kernel_trace_stats.txt | head -16; grep -E "t128" $OUT/prep_pmc.txt | cut -c1-40,70-140
(cd $OUT && rm -rf prep_trace prep_pmc_mfma prep_pmc_lds)
