#!/bin/bash
# SQ counters of the batch-major path's kernels at config 4 (one tiled + one fused run)
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DBS=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d $OUT/tiles_pmc_$name -o t -- python $REPO/tools/gpu_tiles_check.py c4 "$@" > $OUT/tiles_pmc_$name.log 2>&1
  DBS="$DBS $OUT/tiles_pmc_$name/t_results.db"
done
cd $REPO
python tools/rocpd_summary.py pmc $DBS > $OUT/tiles_pmc.txt 2>&1
grep -E "point_pass|pair_tile|rollout_kernel<4, 1024, 4, true, false" $OUT/tiles_pmc.txt | cut -c1-45,70-140
rm -rf $OUT/tiles_pmc_SQ_WAVES $OUT/tiles_pmc_SQ_INSTS_SALU
