import sys
sys.path.insert(0, '.')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, B, thr in [("c2", 256, 0), ("c1", 256, 0), ("c2", 256, 512)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    eng.set_option("threads", thr)
    print(f"== {name} B={B} threads={thr}", flush=True)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 1)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 3)
    torch.cuda.synchronize()
    print(f"   {ms:.3f} ms/launch ({ms*1e3/w.actions.shape[1]:.1f} us/step)", flush=True)
