#!/bin/bash
# SQ counters of the gradient kernels of the final build: config 4 (separable + tile moments) and D = 16 at N = 4096 (wide moment pass)
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
bash tools/gpu_grad_pmc.sh c4 2048 > $OUT/r03_c4_gradient_pmc_head.txt 2>&1
cp $OUT/grad_pmc.txt $OUT/r03_c4_gradient_pmc.txt
cd /tmp && export TMPDIR=/tmp
DBS=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d $OUT/wide_pmc_$name -o t -- python $REPO/tools/gpu_grad_wide_check.py time > $OUT/wide_pmc_$name.log 2>&1
  DBS="$DBS $OUT/wide_pmc_$name/t_results.db"
done
cd $REPO
python tools/rocpd_summary.py pmc $DBS > $OUT/r03_c5class_wide_gradient_pmc.txt 2>&1
grep -E "wide_pair_moments" $OUT/r03_c5class_wide_gradient_pmc.txt | cut -c60-200
grep -E "sep_grad|pair_tile_moments" $OUT/r03_c4_gradient_pmc.txt | cut -c60-200
rm -rf $OUT/wide_pmc_SQ_WAVES $OUT/wide_pmc_SQ_INSTS_SALU
