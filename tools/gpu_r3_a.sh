#!/bin/bash
# Round 3, first GPU visit: batch-major tiles (parity + A/B timing), config-5 kernel trace + counters at N = 4096,
# FETCH_SIZE calibration on known byte counts.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 900 python tools/gpu_tiles_check.py parity time > $OUT/tiles_check.log 2>&1
tail -40 $OUT/tiles_check.log
cd /tmp && export TMPDIR=/tmp
# --- FETCH_SIZE calibration
timeout 60 $REPO/tools/microbench/fetch_calib > $OUT/fetch_calib_run.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_HIT|TCC_MISS|TCC_REQ" | head -40 > $OUT/pmc_counter_list.txt
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch -o calib -- $REPO/tools/microbench/fetch_calib > $OUT/calib_fetch.log 2>&1
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/calib_rdreq -o calib -- $REPO/tools/microbench/fetch_calib > $OUT/calib_rdreq.log 2>&1
# --- config 5 at full N: kernel trace + SQ / LDS counters of rollout_stream_kernel<16, 1024>
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/c5n4096_trace -o c5 -- python $REPO/tools/gpu_c5_step.py 4096:1 4096:2 > $OUT/c5n4096_trace.log 2>&1
DBS=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d $OUT/c5n4096_pmc_$name -o c5 -- python $REPO/tools/gpu_c5_step.py 4096:1 > $OUT/c5n4096_pmc_$name.log 2>&1
  DBS="$DBS $OUT/c5n4096_pmc_$name/c5_results.db"
done
cd $REPO
python tools/rocpd_summary.py pmc $OUT/calib_fetch/calib_results.db $OUT/calib_rdreq/calib_results.db > $OUT/fetch_calib_pmc.txt 2>&1
python tools/rocpd_summary.py trace $OUT/c5n4096_trace/c5_results.db > $OUT/c5n4096_kernel_trace_stats.txt 2>&1
python tools/rocpd_summary.py pmc $DBS > $OUT/c5n4096_pmc.txt 2>&1
cat $OUT/fetch_calib_run.txt; grep read_kernel $OUT/fetch_calib_pmc.txt | cut -c1-40,70-140
head -5 $OUT/c5n4096_kernel_trace_stats.txt | cut -c1-160
grep stream $OUT/c5n4096_pmc.txt | cut -c1-30,70-140
(cd $OUT && rm -rf calib_fetch calib_rdreq c5n4096_trace c5n4096_pmc_SQ_WAVES c5n4096_pmc_SQ_WAIT_INST_LDS)
