#!/bin/bash
# Round 6, evidence run on the shipped build: suite, smoke, counters + bench lines of c1-c4 (c5: one forward line), few-candidate
# latency lines and kernel traces (B = 1), the control step of the default path, gradient kernel traces.  Stamped with gpmpc_build_id().
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
T=${1:-r06z}
python -c "import gp_mpc_amd; print(gp_mpc_amd._lib.lib().gpmpc_build_id().decode())" > $OUT/${T}_build_id.txt 2>/dev/null
timeout 1800 python -m pytest tests -m gpu -q -rf 2>&1 | tail -8 > $OUT/${T}_pytest_gpu_tail.log
tail -2 $OUT/${T}_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/${T}_parity_report.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/${T}_smoke.log
bash tools/gpu_counters.sh ${T}_c2 c2:N200:B256 rollout_kernel --workload c2 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c3 c3:N500:B1024 rollout_kernel --workload c3 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c1 c1:N50:B256 rollout_kernel --workload c1 2>&1 | tail -1
bash tools/gpu_counters.sh ${T}_c4 c4:N1000:B2048 "pair_tile_kernel*30,point_pass_kernel*30,step_params_kernel*30,step_combine_kernel*30" --workload c4 2>&1 | tail -1
# few-candidate lines: the reference's own regime (one candidate per objective evaluation)
PMC_GROUPS="0 1 2 4" bash tools/gpu_counters.sh ${T}_c2_B1 c2:N200:B1 rollout_kernel --workload c2 --candidates-per-gpu 1 2>&1 | tail -1
PMC_GROUPS="0 1 2 4" bash tools/gpu_counters.sh ${T}_c1_B1 c1:N50:B1 rollout_kernel --workload c1 --candidates-per-gpu 1 2>&1 | tail -1
timeout 600 python bench.py > $OUT/${T}_c2_bench.json 2> $OUT/${T}_c2_bench.err
timeout 600 python bench.py --workload c3 > $OUT/${T}_c3_bench.json 2> $OUT/${T}_c3_bench.err
timeout 600 python bench.py --workload c1 > $OUT/${T}_c1_bench.json 2> $OUT/${T}_c1_bench.err
timeout 900 python bench.py --workload c4 > $OUT/${T}_c4_bench.json 2> $OUT/${T}_c4_bench.err
timeout 300 python bench.py --no-cpu-baseline --candidates-per-gpu 4096 > $OUT/${T}_c2_B4096_bench.json 2> $OUT/${T}_c2_B4096_bench.err
timeout 300 python bench.py --workload c2 --candidates-per-gpu 1 --no-cpu-baseline > $OUT/${T}_c2_B1_bench.json 2> $OUT/${T}_c2_B1_bench.err
timeout 300 python bench.py --workload c1 --candidates-per-gpu 1 --no-cpu-baseline > $OUT/${T}_c1_B1_bench.json 2> $OUT/${T}_c1_B1_bench.err
timeout 300 python bench.py --workload c3 --candidates-per-gpu 1 --no-cpu-baseline > $OUT/${T}_c3_B1_bench.json 2> $OUT/${T}_c3_B1_bench.err
timeout 300 python bench.py --steps 2000 --no-cpu-baseline > $OUT/${T}_c2_bench_steps2000.json 2> /dev/null
timeout 300 python bench.py --force-dist --no-cpu-baseline > $OUT/${T}_c2_bench_dist_world1.json 2> /dev/null
timeout 300 python tools/gpu_control_step.py 2>&1 | grep -v amdgpu > $OUT/${T}_control_step.txt
timeout 600 python tools/gpu_cluster_sweep.py c2,c3,c1,n100 1 2>&1 | grep -v amdgpu > $OUT/${T}_cluster_sweep.txt
for b in 16 64 128; do timeout 300 python tools/gpu_cluster_sweep.py c2,c3 $b 2>&1 | grep -v amdgpu >> $OUT/${T}_cluster_sweep.txt; done
cd /tmp && export TMPDIR=/tmp
F=$OUT/${T}_b1_latency.txt; : > $F
for cs in 1 0; do
  for wl in c2 c3; do
    timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/b1_$cs -o g -- python $REPO/tools/gpu_grad_profile.py $wl 1 20 cluster=$cs > $OUT/b1_$cs.log 2>&1
    echo "== $wl, B = 1, option cluster = $cs (1: one workgroup per candidate; 0: default dispatch): wall clock per launch (20 launches), then the kernel trace" >> $F
    grep -a "ms per launch" $OUT/b1_$cs.log >> $F
    (cd $REPO && python tools/rocpd_summary.py trace $OUT/b1_$cs/g_results.db | head -9 | cut -c1-150) >> $F
    rm -rf $OUT/b1_$cs $OUT/b1_$cs.log
  done
done
for wl in "c2 256" "c4 2048"; do
  set -- $wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${T}_g$1 -o g -- python $REPO/tools/gpu_grad_profile.py $1 $2 > $OUT/${T}_g$1.log 2>&1
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/${T}_g$1/g_results.db > $OUT/${T}_$1_gradient_kernel_trace_stats.txt 2>&1)
  rm -rf $OUT/${T}_g$1
done
cd $REPO
if [ -f gpurun_dbg/libgpmpc_hip_prof.so ]; then
  for o in "cluster=1" "cluster=0"; do
    echo "== prof build, c2 B = 1, option $o: cycles per phase summed over the horizon (workgroup 0); wall clock per member" >> $F
    GPMPC_LIB=$REPO/gpurun_dbg/libgpmpc_hip_prof.so timeout 120 python tools/gpu_grad_profile.py c2 1 1 $o 2>&1 | grep -a "PROF cycles\|PROF sweep\|PROF member  [0-3] wall" | sort | uniq | head -8 >> $F
  done
fi
if [ -z "$SKIP_C5" ]; then
  timeout 1500 python bench.py --workload c5 --candidates-total 256 --steps 1 --warmup 0 --no-gradient > $OUT/${T}_c5_bench_B256.json 2> $OUT/${T}_c5_bench.err
fi
bash tools/gpu_gram_ab.sh > /dev/null 2>&1; cp $OUT/r06_gram_ab.txt $OUT/${T}_gram_ab.txt
tail -2 $OUT/${T}_pytest_gpu_tail.log; cat $OUT/${T}_smoke.log; cat $OUT/${T}_control_step.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("gpurun_out/${T}_c*_bench*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f ms/step %.4f closed %s kernel_ms %.4f frac %.3f form %.3f valu_busy %s exec %s grad_ms %s host_eval %s prepare %.3f wg/cand %s note %s" % (
            d["value"], d["ms_per_step"], d.get("closed_loop_ms_per_step"), r["kernel_ms"], r["frac"], r["formulation"]["frac_formulation"], r["valu_busy_frac"], (r["executed"] or {}).get("frac_of_peak"), g.get("ms_per_launch"), g.get("host_in_host_out_ms_per_evaluation"), d["prepare_ms"], d["config"].get("workgroups_per_candidate"), r["counters_note"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
