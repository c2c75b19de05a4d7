"""Forward rollout time per launch at B = 1 and at the bench batch, c1 / c2 / c3 (default dispatch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, Bs in (("c2", (1, 256)), ("c3", (1, 1024)), ("c1", (1, 256))):
    n, d, a, h, b, tm = synth.SHAPES[name]
    w = synth.make_workload(n, d, a, h, max(Bs), include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    for B in Bs:
        acts = torch.as_tensor(w.actions[:B], device="cuda:0")
        out = eng.rollout(acts, w.mu0, w.S0)
        for _ in range(200): eng.rollout(acts, w.mu0, w.S0, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): eng.rollout(acts, w.mu0, w.S0, out=out)
        torch.cuda.synchronize()
        print(f"{name} B={B}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per launch (cluster {eng.last_cluster})", flush=True)
eng.close()
