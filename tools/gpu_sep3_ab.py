"""D = 3 at larger N: compile-time-structured separable moments (exact-D kernel) vs the generic blocked form (runtime-D kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for N in (200, 400, 800):
    w = synth.make_workload(N, 3, 1, 20, 256, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    eng.rollout_timed(acts, w.mu0, w.S0, 30 if N < 500 else 8)
    best = {}
    for rep in range(3):
        for name, ed, fp in (("static moments", 0, 0), ("blocked moments (runtime D)", 2, 0), ("element-wise", 0, 2)):
            eng.set_option("exact_dim", ed); eng.set_option("force_path", fp)
            ms, J = eng.rollout_timed(acts, w.mu0, w.S0, 10 if N < 500 else 4)
            best[name] = min(best.get(name, 1e9), ms)
    eng.set_option("exact_dim", 0); eng.set_option("force_path", 0)
    print(f"N={N} D=3 H=20 B=256: " + "  ".join(f"{k}: {v:.3f} ms" for k, v in best.items()), flush=True)
eng.close()
