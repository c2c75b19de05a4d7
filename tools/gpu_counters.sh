#!/bin/bash
# rocprofv3 kernel trace + PMC passes (each counter group in its own run, no trace domains mixed in) of one bench
# workload; summaries -> gpurun_out/<tag>_*.txt, per-launch averages merged into profiles/pmc_counters.json and
# profiles/pmc_traffic.json under the key bench.py looks up.
#   tools/gpu_counters.sh <tag> <key> <kernel-name-substring> <bench args...>
#   e.g. tools/gpu_counters.sh r02_c2 c2:N200:B256 rollout_kernel --workload c2
#        tools/gpu_counters.sh r03_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30" --workload c4
#   (a launch made of several kernels per horizon step: sum of average-per-dispatch x count)
TAG=$1; KEY=$2; KSUB=$3; shift 3
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline $@"
# launches per run: TRACE_RUN / PMC_RUN override them (config 5: one 28 s launch per pass)
TRACE_RUN=${TRACE_RUN:---steps 20 --warmup 3}
PMC_RUN=${PMC_RUN:---steps 3 --warmup 1}
PASS_LIMIT=${PASS_LIMIT:-300}
[ -z "$SKIP_TRACE" ] && timeout $PASS_LIMIT rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o bench -- $B $TRACE_RUN > $OUT/${TAG}_trace.log 2>&1
DBS=""
GI=-1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE"; do
  GI=$((GI + 1))
  # PMC_GROUPS: indices of the counter groups to collect (default: all five)
  if [ -n "$PMC_GROUPS" ] && ! echo " $PMC_GROUPS " | grep -q " $GI "; then continue; fi
  name=$(echo $grp | cut -d' ' -f1)
  timeout $PASS_LIMIT rocprofv3 --pmc $grp -d $OUT/${TAG}_pmc_$name -o bench -- $B $PMC_RUN > $OUT/${TAG}_pmc_$name.log 2>&1
  DBS="$DBS $OUT/${TAG}_pmc_$name/bench_results.db"
done
cd $REPO
[ -z "$SKIP_TRACE" ] && python tools/rocpd_summary.py trace $OUT/${TAG}_trace/bench_results.db > $OUT/${TAG}_kernel_trace_stats.txt
python tools/rocpd_summary.py pmc $DBS > $OUT/${TAG}_pmc.txt
BID=$(python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())")
python tools/rocpd_summary.py json --build-id=$BID $KEY "$KSUB" $DBS | cut -c1-300
cp profiles/pmc_counters.json profiles/pmc_traffic.json $OUT/
[ -z "$SKIP_TRACE" ] && head -6 $OUT/${TAG}_kernel_trace_stats.txt | cut -c1-150
# the sqlite outputs are tens of MB each: only the text summaries travel back (gpurun merges at most 64 MiB)
(cd $OUT && rm -rf ${TAG}_trace ${TAG}_pmc_*)
