import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
from oracle import gpmpc_oracle as orc
from helpers import rel_err
eng = gp_mpc_amd.HipEngine(0)
for (N, D, A) in [(200, 3, 1), (500, 2, 1), (1000, 4, 2), (2048, 4, 2), (4096, 16, 4)]:
    w = synth.make_workload(N, D, A, 2, 2, seed=1)
    X, Y = torch.as_tensor(w.X).cuda(), torch.as_tensor(w.Y).cuda()
    ls, osc, nz = torch.as_tensor(w.lengthscales).cuda(), torch.as_tensor(w.outputscales).cuda(), torch.as_tensor(w.noises).cuda()
    eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.prepare(X, Y, ls, osc, nz); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    msg = f"prepare N={N} D={D}: {min(ts)*1e3:.2f} ms"
    flops = D * (N**3 / 3 + N**3 / 3 + N**3 / 3)
    msg += f"  ({flops / min(ts) / 1e12:.2f} TFLOP/s on the N^3 contractions)"
    if N <= 2048:
        iK, beta = eng.factors()
        iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        msg += f"  rel err iK {rel_err(iK.cpu().numpy(), iK0):.1e} beta {rel_err(beta.cpu().numpy(), beta0):.1e}"
    print(msg, flush=True)
