#!/bin/bash
# Round 5, second evidence run: the shipped build after the 48-row chunk cap of the two-column form (host-side launch rule; the
# config-5 kernels are not touched: their line and counters stay those of tools/gpu_r5_final.sh).  Suite, smoke, counters + bench
# lines of c1 - c4 and of c2 at 4096 candidates per GPU.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=r05y
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/${T}_build_id.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -8 > $OUT/${T}_pytest_gpu_tail.log
tail -2 $OUT/${T}_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/${T}_parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/${T}_smoke.log
bash tools/gpu_counters.sh ${T}_c2 c2:N200:B256 rollout_kernel --workload c2 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c3 c3:N500:B1024 rollout_kernel --workload c3 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c1 c1:N50:B256 rollout_kernel --workload c1 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30,step_combine_kernel*30" --workload c4 2>&1 | tail -1
SKIP_TRACE=1 bash tools/gpu_counters.sh ${T}_c2_B4096 c2:N200:B4096 rollout_kernel --workload c2 --candidates-per-gpu 4096 2>&1 | tail -1
timeout 600 python bench.py > $OUT/${T}_c2_bench.json 2> $OUT/${T}_c2_bench.err
timeout 600 python bench.py --workload c3 > $OUT/${T}_c3_bench.json 2> $OUT/${T}_c3_bench.err
timeout 600 python bench.py --workload c1 > $OUT/${T}_c1_bench.json 2> $OUT/${T}_c1_bench.err
timeout 900 python bench.py --workload c4 > $OUT/${T}_c4_bench.json 2> $OUT/${T}_c4_bench.err
timeout 300 python bench.py --no-cpu-baseline --candidates-per-gpu 4096 > $OUT/${T}_c2_B4096_bench.json 2> $OUT/${T}_c2_B4096_bench.err
cat $OUT/${T}_smoke.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05y_c*_bench*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f frac %.3f valu_busy %s exec %s grad_ms %s prepare %.3f spread %.4f note %s" % (
            d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["valu_busy_frac"], (r["executed"] or {}).get("frac_of_peak"), g.get("ms_per_launch"), d["prepare_ms"],
            d["windows"]["spread"], r["counters_note"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
