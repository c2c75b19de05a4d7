#!/bin/bash
# Round 3, after the last kernel change (wide-gradient moment pass): GPU suite, smoke, counters of c2 / c4 on the new build id,
# their bench lines, the config-5 bench line.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/r03_build_id.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/r03_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r03_parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/r03_smoke.log
bash tools/gpu_counters.sh r03_c2 c2:N200:B256 rollout_kernel --workload c2 2>&1 | tail -3
bash tools/gpu_counters.sh r03_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30,step_combine_kernel*30" --workload c4 2>&1 | tail -3
timeout 600 python bench.py > $OUT/r03_c2_bench.json 2> $OUT/r03_c2_bench.err
timeout 900 python bench.py --workload c4 > $OUT/r03_c4_bench.json 2> $OUT/r03_c4_bench.err
timeout 1100 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 > $OUT/r03_c5_bench_B256.json 2> $OUT/r03_c5_bench.err
tail -2 $OUT/r03_pytest_gpu_tail.log; cat $OUT/r03_smoke.log; for f in $OUT/r03_c2_bench.json $OUT/r03_c4_bench.json $OUT/r03_c5_bench_B256.json; do cut -c1-200 $f; done
