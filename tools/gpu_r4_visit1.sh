#!/bin/bash
# Round 4, first GPU visit: suite + smoke on the new build, counters of c1 / c3 and of c2 at B = 1024 / 4096 per GPU, their bench
# lines, the late-horizon config-5 step times + LDS counters (A/B over the exponential forms), the config-5 bench line with parity.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/r04a_build_id.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -25 > $OUT/r04a_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04a_parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/r04a_smoke.log
bash tools/gpu_counters.sh r04_c1 c1:N50:B256 rollout_kernel --workload c1 2>&1 | tail -2
bash tools/gpu_counters.sh r04_c3 c3:N500:B1024 rollout_kernel --workload c3 2>&1 | tail -2
bash tools/gpu_counters.sh r04_c2_B1024 c2:N200:B1024 rollout_kernel --workload c2 --candidates-per-gpu 1024 2>&1 | tail -2
bash tools/gpu_counters.sh r04_c2_B4096 c2:N200:B4096 rollout_kernel --workload c2 --candidates-per-gpu 4096 2>&1 | tail -2
timeout 300 python bench.py --workload c1 > $OUT/r04a_c1_bench.json 2> $OUT/r04a_c1_bench.err
timeout 300 python bench.py --workload c3 > $OUT/r04a_c3_bench.json 2> $OUT/r04a_c3_bench.err
timeout 300 python bench.py --no-cpu-baseline --candidates-per-gpu 1024 > $OUT/r04a_c2_B1024_bench.json 2> $OUT/r04a_c2_B1024_bench.err
timeout 300 python bench.py --no-cpu-baseline --candidates-per-gpu 4096 > $OUT/r04a_c2_B4096_bench.json 2> $OUT/r04a_c2_B4096_bench.err
timeout 300 python bench.py > $OUT/r04a_c2_bench.json 2> $OUT/r04a_c2_bench.err
# config 5: late-horizon states
timeout 300 python tools/gpu_c5_late.py 2>&1 | grep "state of" | tee $OUT/r04_c5_late_horizon_step_times.txt
cd /tmp && export TMPDIR=/tmp
for tag in "t0:0:force_path=0" "t25:25:force_path=0" "t25_direct:25:force_path=1" "t0_tab:0:force_path=4"; do
  name=$(echo $tag | cut -d: -f1); t=$(echo $tag | cut -d: -f2); opt=$(echo $tag | cut -d: -f3)
  DBS=""
  for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
             "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH"; do
    g=$(echo $grp | cut -d' ' -f1)
    timeout 150 rocprofv3 --pmc $grp -d $OUT/c5l_${name}_$g -o c5 -- python $REPO/tools/gpu_c5_late.py $t $opt > $OUT/c5l_${name}_$g.log 2>&1
    DBS="$DBS $OUT/c5l_${name}_$g/c5_results.db"
  done
  (cd $REPO && python tools/rocpd_summary.py pmc $DBS | grep -E "stream|^#|kernel " > $OUT/r04_c5_late_pmc_$name.txt)
  (cd $OUT && rm -rf c5l_${name}_SQ_WAVE_CYCLES c5l_${name}_SQ_WAIT_INST_LDS c5l_${name}_SQ_INSTS_LDS_LOAD)
done
cd $REPO
timeout 900 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 --no-gradient > $OUT/r04a_c5_bench_B256_nograd.json 2> $OUT/r04a_c5_bench.err
tail -4 $OUT/r04a_pytest_gpu_tail.log; cat $OUT/r04a_smoke.log; for f in $OUT/r04a_c*_bench*.json; do echo $f; cut -c1-150 $f; done
