#!/bin/bash
# Round 4, fourth GPU visit: suite on the build with the mean-moments kernel and the two-stream prepare; prepare A/B at N = 300 .. 600
# (option prepare_overlap), config-4 bench (gradient with the mean kernel), LDS counters of the config-5 kernel with the odd stage stride.
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $OUT/r04d_pytest_gpu_tail.log
cp $OUT/parity_report.json $OUT/r04d_parity_report.json 2>/dev/null
tail -4 $OUT/r04d_pytest_gpu_tail.log
timeout 300 python - > $OUT/r04d_prepare_overlap_ab.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
eng.set_option("incremental", 0)
for (N, D, A) in [(300, 3, 1), (400, 4, 2), (500, 2, 1), (500, 4, 2), (600, 4, 2)]:
    w = synth.make_workload(N, D, A, 2, 2, seed=1)
    X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
    ls, osc, nz = (torch.as_tensor(v).cuda() for v in (w.lengthscales, w.outputscales, w.noises))
    res = {}
    for ov in (1, 0, 1, 0):
        eng.set_option("prepare_overlap", ov)
        eng.prepare(X, Y, ls, osc, nz)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.setdefault(ov, []).append(float(np.median(ts)) * 1e3)
        if ov == 1:
            iK1 = eng.factors()[0].clone()
        else:
            assert torch.equal(iK1, eng.factors()[0]), "the two-stream prepare must give the same factors bit for bit"
    print(f"N={N} D={D}: prepare two streams {min(res[1]):.3f} ms, one stream {min(res[0]):.3f} ms (factors identical)", flush=True)
eng.set_option("prepare_overlap", 1)
PY
cat $OUT/r04d_prepare_overlap_ab.txt
timeout 400 python bench.py --no-cpu-baseline --workload c4 > $OUT/r04d_c4.json 2> $OUT/r04d_c4.err
timeout 400 python bench.py --no-cpu-baseline --workload c4 --option grad_mean=0 > $OUT/r04d_c4_mean_inpass.json 2> $OUT/r04d_c4_mean_inpass.err
timeout 300 python bench.py --no-cpu-baseline --workload c3 > $OUT/r04d_c3.json 2> $OUT/r04d_c3.err
cd /tmp && export TMPDIR=/tmp
DBS=""
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  g=$(echo $grp | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $grp -d $OUT/c5o_$g -o c5 -- python $REPO/tools/gpu_c5_late.py 0 25 > $OUT/c5o_$g.log 2>&1
  DBS="$DBS $OUT/c5o_$g/c5_results.db"
done
(cd $REPO && python tools/rocpd_summary.py pmc $DBS | grep -E "stream|^#|kernel " > $OUT/r04d_c5_odd_stride_pmc.txt)
(cd $OUT && rm -rf c5o_SQ_WAVE_CYCLES c5o_SQ_WAIT_INST_LDS)
cd $REPO
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r04d_c*.json")):
    try:
        d = json.load(open(f))
        g = d.get("gradient") or {}
        print(os.path.basename(f), "value %.1f ms/step %.4f kernel_ms %.4f grad_ms %s prepare %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], g.get("ms_per_launch"), d["prepare_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
grep -E "BANK_CONFLICT|IDX_ACTIVE|SQ_INSTS_LDS " $OUT/r04d_c5_odd_stride_pmc.txt | cut -c1-40,70-160
