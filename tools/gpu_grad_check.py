"""Analytic-gradient kernels vs the numpy adjoint (oracle/adjoint.py) and vs finite differences (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import gp_mpc_amd
from oracle import synth, gpmpc_oracle as orc, adjoint

eng = gp_mpc_amd.HipEngine(0)
for (N, D, A, H, B, tm) in [(30, 3, 1, 5, 3, False), (25, 2, 2, 4, 2, True), (70, 4, 2, 3, 2, False), (200, 3, 1, 25, 4, False),
                            (90, 6, 2, 4, 2, True), (40, 8, 3, 3, 2, False)]:
    w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=1)
    f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    out = eng.rollout_grad(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    torch.cuda.synchronize()
    g = out["grad"].cpu().numpy()
    worst = 0.0
    for b in range(min(B, 2)):
        J, gr, *_ = adjoint.lcb_and_gradient(f, w.actions[b], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa, w.include_time, w.time0)
        worst = max(worst, np.abs(g[b] - gr).max() / np.abs(gr).max())
        jerr = abs(float(out["J"][b]) - J) / abs(J)
    t0 = time.perf_counter()
    for _ in range(5):
        out = eng.rollout_grad(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        o2 = eng.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / 5
    print(f"N={N} D={D} A={A} H={H} B={B} time={tm}: grad rel err {worst:.2e}  J rel err {jerr:.2e}  grad launch {tg*1e3:.3f} ms  forward {tf*1e3:.3f} ms", flush=True)
eng.close()
