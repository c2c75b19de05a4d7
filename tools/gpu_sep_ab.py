"""A/B of the element-wise vs the separable (moment) evaluation of the off-diagonal pairs (HIP-event kernel time)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for shape, B in (("c2", 256), ("c1", 256), ("c3", 256), ("c4", 128)):
    n, d, a, h, b, tm = synth.SHAPES[shape]
    w = synth.make_workload(n, d, a, h, B, include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    line = f"{shape} N={n} D={d} H={h} B={B}:"
    modes = (("auto", 0), ("separable forced", 1))
    best = {m[0]: 1e9 for m in modes}
    eng.rollout_timed(acts, w.mu0, w.S0, 60 if n < 600 else 5, w.include_time, w.time0)       # clocks
    for rep in range(4):                                     # interleaved, best of 4: the clocks wander by several percent
        for name, fs in modes:
            eng.set_option("force_separable", fs)
            ms, J = eng.rollout_timed(acts, w.mu0, w.S0, 20 if n < 600 else 3, w.include_time, w.time0)
            best[name] = min(best[name], ms)
    eng.set_option("force_separable", 0)
    print(line + "".join(f"  {k}: {v:.3f} ms" for k, v in best.items()), flush=True)
eng.close()
