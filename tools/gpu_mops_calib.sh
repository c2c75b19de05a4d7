#!/bin/bash
# Unit of SQ_INSTS_VALU_MFMA_MOPS_F64: the fp64 matrix-instruction probe (tools/microbench/mfma_f64_rate: only v_mfma_f64_16x16x4_f64,
# 1024 multiply-adds each) under the counters -- MOPS per SQ_INSTS_MFMA = counts per instruction.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/mops -o m -- $REPO/tools/microbench/mfma_f64_rate > $OUT/mops.log 2>&1
cd $REPO
python tools/rocpd_summary.py pmc $OUT/mops/m_results.db > $OUT/r04z_mfma_mops_unit.txt 2>&1
rm -rf $OUT/mops
cat $OUT/r04z_mfma_mops_unit.txt | cut -c1-160
tail -5 $OUT/mops.log | cut -c1-200
