#!/bin/bash
# Round 3, final evidence on the shipped build: GPU suite, smoke, counters (c2 fused kernel, c4 batch-major path), bench lines of
# all BASELINE shapes, gradient kernel traces.  Everything lands in gpurun_out/ (copied to profiles/ by hand afterwards).
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
for w in c1 c3 c4; do timeout 900 python bench.py --workload $w > $OUT/r03_${w}_bench.json 2> $OUT/r03_${w}_bench.err; done
timeout 900 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 > $OUT/r03_c5_bench_B256.json 2> $OUT/r03_c5_bench.err
bash tools/gpu_grad_prof.sh c2 256 > $OUT/r03_c2_gradient.log 2>&1; cp $OUT/grad_kernel_trace_stats.txt $OUT/r03_c2_gradient_kernel_trace_stats.txt
bash tools/gpu_grad_prof.sh c4 2048 > $OUT/r03_c4_gradient.log 2>&1; cp $OUT/grad_kernel_trace_stats.txt $OUT/r03_c4_gradient_kernel_trace_stats.txt
rm -rf $OUT/grad_trace
tail -2 $OUT/r03_pytest_gpu_tail.log; cat $OUT/r03_smoke.log; for f in $OUT/r03_c*_bench*.json; do cut -c1-260 $f; done
