#!/bin/bash
# Round 3: counters of the shipped build for c2 (fused kernel) and c4 (batch-major path), exchange test
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_VALU_MFMA[A-Z0-9_]*" | sort -u > $OUT/pmc_valu_counter_names.txt
timeout 600 python -m pytest tests/test_gpu_controller.py -x -q -k "exchange or process_group" -s 2>&1 | grep -E "ms/step|passed|failed|Error" | tee $OUT/exchange_test.log
bash tools/gpu_counters.sh r03_c2 c2:N200:B256 rollout_kernel --workload c2 2>&1 | tail -4
bash tools/gpu_counters.sh r03_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30" --workload c4 2>&1 | tail -4
