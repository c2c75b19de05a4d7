#!/bin/bash
# Round 5, evidence of config 5 on the final build (tools/gpu_r5_final2.sh covered c1 - c4): the four counter passes and the bench
# line with its gradient leg.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
PMC_GROUPS="0 1 2 4" SKIP_TRACE=1 PMC_RUN="--steps 1 --warmup 0 --no-gradient --no-batch-check" PASS_LIMIT=400 bash tools/gpu_counters.sh r05y_c5 c5:N4096:B256 rollout_stream_kernel --workload c5 --candidates-total 256 2>&1 | tail -1
timeout 1500 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 > $OUT/r05y_c5_bench_B256.json 2> $OUT/r05y_c5_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05y_c5_bench_B256.json")); r = d["roofline"]
print("c5 value %.3f ms/step %.1f kernel_ms %.1f frac %.3f valu_busy %s exec %s grad_ms %s prepare %.2f note %s batch %s parity %s" % (
    d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["valu_busy_frac"], (r["executed"] or {}).get("frac_of_peak"),
    d["gradient"]["ms_per_launch"], d["prepare_ms"], r["counters_note"], d["parity"].get("batch_independence"), {k: v for k, v in d["parity"].items() if k.startswith("max")}))
PY
