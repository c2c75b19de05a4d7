"""Time of one training-loss + gradient evaluation (gpmpc_mll) vs the CPU torch autograd expression."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import gp_mpc_amd
from oracle import synth, gp_training

eng = gp_mpc_amd.HipEngine(0)
for (N, D, A) in ((200, 3, 1), (500, 2, 1), (1000, 4, 2)):
    w = synth.make_workload(N, D, A, 3, 2, seed=N)
    X = torch.as_tensor(w.X, device="cuda:0"); Y = torch.as_tensor(w.Y, device="cuda:0")
    eng.mll(X, Y, w.lengthscales, w.outputscales, w.noises)
    t0 = time.perf_counter()
    for _ in range(10):
        eng.mll(X, Y, w.lengthscales, w.outputscales, w.noises)
    tg = (time.perf_counter() - t0) / 10
    Xc, tt = torch.as_tensor(w.X), lambda v: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True)
    torch.set_num_threads(8)
    t0 = time.perf_counter()
    for a in range(D):
        ls, osc, nz = tt(w.lengthscales[a]), tt(w.outputscales[a]), tt(w.noises[a])
        gp_training.neg_mll_torch(Xc, torch.as_tensor(w.Y[:, a]), ls, osc, nz).backward()
    tc = time.perf_counter() - t0
    print(f"N={N} D={D}: gpmpc_mll (all {D} GPs, loss + gradient) {tg*1e3:.3f} ms; CPU torch autograd (8 threads) {tc*1e3:.1f} ms", flush=True)
eng.close()
