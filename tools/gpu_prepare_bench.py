"""gpmpc_prepare: full factorisation time (reuse switched off) and accuracy vs the CPU oracle.
  python tools/gpu_prepare_bench.py [N:D:A ...]     N <= 240 is timed with the fused single-launch factorisation and with the panel path"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
from oracle import gpmpc_oracle as orc
from helpers import rel_err
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(50, 3, 1), (200, 3, 1), (256, 4, 2), (500, 2, 1), (1000, 4, 2), (4096, 16, 4)]
eng = gp_mpc_amd.HipEngine(0)
eng.set_option("incremental", 0)
for (N, D, A) in shapes:
    w = synth.make_workload(N, D, A, 2, 2, seed=1)
    X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
    ls, osc, nz = torch.as_tensor(w.lengthscales).cuda(), torch.as_tensor(w.outputscales).cuda(), torch.as_tensor(w.noises).cuda()
    for fused in ((1, 2, 0) if N <= 240 else (0,)):
        eng.set_option("fused_prepare", fused)
        eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        msg = f"prepare N={N} D={D} { {0: 'panel path', 1: 'default (<= 240: fused factorisation, < 96: all fused)', 2: 'all fused'}[fused] }: {np.median(ts)*1e3:.3f} ms (min {min(ts)*1e3:.3f})"
        flops = D * N**3                        # N^3/3 each: Cholesky, triangular inverse, Y^T Y
        msg += f"  ({flops / np.median(ts) / 1e12:.2f} TFLOP/s on the N^3 contractions)"
        if N <= 2048:
            iK, beta = eng.factors()
            iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
            msg += f"  rel err iK {rel_err(iK.cpu().numpy(), iK0):.1e} beta {rel_err(beta.cpu().numpy(), beta0):.1e}"
        print(msg, flush=True)
    eng.set_option("fused_prepare", 1)
