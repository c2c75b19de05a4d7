#!/bin/bash
# Round 5, evidence run on the shipped build: suite, smoke, counters + bench lines of every BASELINE shape (c5 with its gradient leg),
# gradient kernel traces of c2 / c4.  Everything is stamped with gpmpc_build_id().
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/r05z_build_id.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -8 > $OUT/r05z_pytest_gpu_tail.log
tail -2 $OUT/r05z_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r05z_parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/r05z_smoke.log
bash tools/gpu_counters.sh r05z_c2 c2:N200:B256 rollout_kernel --workload c2 2>&1 | tail -1
bash tools/gpu_counters.sh r05z_c1 c1:N50:B256 rollout_kernel --workload c1 2>&1 | tail -1
bash tools/gpu_counters.sh r05z_c3 c3:N500:B1024 rollout_kernel --workload c3 2>&1 | tail -1
bash tools/gpu_counters.sh r05z_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30,step_combine_kernel*30" --workload c4 2>&1 | tail -1
# config 5: one 28 s launch per step -- the four passes the bench line's counter figures need (traffic, VALU, fp64 / matrix ops), three launches each
PMC_GROUPS="0 1 2 4" SKIP_TRACE=1 PMC_RUN="--steps 1 --warmup 0 --no-gradient" PASS_LIMIT=400 bash tools/gpu_counters.sh r05z_c5 c5:N4096:B256 rollout_stream_kernel --workload c5 --candidates-total 256 2>&1 | tail -1
timeout 600 python bench.py > $OUT/r05z_c2_bench.json 2> $OUT/r05z_c2_bench.err
timeout 600 python bench.py --workload c1 > $OUT/r05z_c1_bench.json 2> $OUT/r05z_c1_bench.err
timeout 600 python bench.py --workload c3 > $OUT/r05z_c3_bench.json 2> $OUT/r05z_c3_bench.err
timeout 900 python bench.py --workload c4 > $OUT/r05z_c4_bench.json 2> $OUT/r05z_c4_bench.err
timeout 300 python bench.py --no-cpu-baseline --candidates-per-gpu 4096 > $OUT/r05z_c2_B4096_bench.json 2> $OUT/r05z_c2_B4096_bench.err
cd /tmp && export TMPDIR=/tmp
for wl in "c2 256" "c4 2048"; do
  set -- $wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r05z_g$1 -o g -- python $REPO/tools/gpu_grad_profile.py $1 $2 > $OUT/r05z_g$1.log 2>&1
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/r05z_g$1/g_results.db > $OUT/r05z_$1_gradient_kernel_trace_stats.txt 2>&1)
  rm -rf $OUT/r05z_g$1
done
cd $REPO
timeout 1500 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 > $OUT/r05z_c5_bench_B256.json 2> $OUT/r05z_c5_bench.err
bash tools/gpu_b1_latency.sh > /dev/null 2>&1; mv $OUT/r04z_b1_latency.txt $OUT/r05z_b1_latency.txt 2>/dev/null
tail -2 $OUT/r05z_pytest_gpu_tail.log; cat $OUT/r05z_smoke.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05z_c*_bench*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f frac %.3f valu_busy %s exec %s grad_ms %s prepare %.3f parity %s note %s" % (
            d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["valu_busy_frac"], (r["executed"] or {}).get("frac_of_peak"), g.get("ms_per_launch"), d["prepare_ms"],
            {k: v for k, v in d["parity"].items() if k.startswith("max")}, r["counters_note"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
