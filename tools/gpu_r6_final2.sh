#!/bin/bash
# Round 6, evidence run part 2 (same build): the control step of the default path, counters of c2 at 4096 candidates per GPU and of
# config 5 (four passes of one 28 s launch each), their bench lines refreshed from the counters.
REPO=$PWD; OUT=$REPO/gpurun_out; T=${1:-r06z}
timeout 300 python tools/gpu_control_step.py 2>&1 | grep -v amdgpu > $OUT/${T}_control_step.txt
cat $OUT/${T}_control_step.txt
SKIP_TRACE=1 bash tools/gpu_counters.sh ${T}_c2_B4096 c2:N200:B4096 rollout_kernel --workload c2 --candidates-per-gpu 4096 2>&1 | tail -1
PMC_GROUPS="0 1 2 4" SKIP_TRACE=1 PMC_RUN="--steps 1 --warmup 0 --no-gradient --no-batch-check" PASS_LIMIT=400 bash tools/gpu_counters.sh ${T}_c5 c5:N4096:B256 rollout_stream_kernel --workload c5 --candidates-total 256 2>&1 | tail -1
cp profiles/pmc_counters.json profiles/pmc_traffic.json $OUT/
