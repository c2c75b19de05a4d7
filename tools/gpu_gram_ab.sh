#!/bin/bash
# K build at large N: shared-difference kernel (gram_lower_kernel, all GPs per tile) vs the per-GP kernel -- kernel trace of gpmpc_prepare
# at config 5 (N = 4096, D = 16, E = 20) and N = 1000 (D = 4, E = 6); then WRITE_SIZE of both (PMC pass)
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
cat > /tmp/gram_run.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
import gp_mpc_amd
from oracle import synth
opt = int(sys.argv[2])
eng = gp_mpc_amd.HipEngine(0)
eng.set_option("incremental", 0); eng.set_option("gram_shared", opt)
for (N, D, A) in ((4096, 16, 4), (1000, 4, 2)):
    w = synth.make_workload(N, D, A, 2, 2, seed=1)
    t = lambda a: torch.as_tensor(a).cuda()
    X, Y, ls, osc, nz = t(w.X), t(w.Y), t(w.lengthscales), t(w.outputscales), t(w.noises)
    for _ in range(4):
        eng.prepare(X, Y, ls, osc, nz)
    torch.cuda.synchronize()
    iK, beta = eng.factors()
    print("N", N, "gram_shared", opt, "beta checksum %.15e" % float(beta.abs().sum()), "iK checksum %.15e" % float(iK.abs().sum()))
PY
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06_gram_ab.txt; : > $F
for o in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/gram_$o -o g -- python /tmp/gram_run.py $REPO $o > $OUT/gram_$o.log 2>&1
  echo "== option gram_shared = $o" >> $F; grep -a "checksum" $OUT/gram_$o.log >> $F
  (cd $REPO && python tools/rocpd_summary.py trace $OUT/gram_$o/g_results.db | grep -i "gram\|kernel  " | cut -c1-150) >> $F
  rm -rf $OUT/gram_$o
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/gramp_$o -o g -- python /tmp/gram_run.py $REPO $o > /dev/null 2>&1
  (cd $REPO && python tools/rocpd_summary.py pmc $OUT/gramp_$o/g_results.db 2>/dev/null | grep -i "gram" | cut -c1-200) >> $F
  rm -rf $OUT/gramp_$o
done
cat $F
