#!/bin/bash
# SQ counters of the gradient path's kernels (args: shape B [engine options])
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
SHAPE=${1:-c4}; B=${2:-2048}; shift; shift
cd /tmp && export TMPDIR=/tmp
DBS=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d $OUT/grad_pmc_$name -o t -- python $REPO/tools/gpu_grad_profile.py $SHAPE $B 1 "$@" > $OUT/grad_pmc_$name.log 2>&1
  DBS="$DBS $OUT/grad_pmc_$name/t_results.db"
done
cd $REPO
python tools/rocpd_summary.py pmc $DBS > $OUT/grad_pmc.txt 2>&1
grep -E "^kernel|sep_grad|pair_tile_moments|pair_moments" $OUT/grad_pmc.txt | cut -c1-60,100-260
rm -rf $OUT/grad_pmc_SQ_WAVES $OUT/grad_pmc_SQ_INSTS_SALU
