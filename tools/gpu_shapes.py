"""Kernel time of the forward rollout at the BASELINE shapes c1..c4 (B = 256, default options), best of 3 x 5 launches."""
import sys
sys.path.insert(0, '.')
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for name, B in [("c1", 256), ("c2", 256), ("c3", 256), ("c4", 128)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    eng.rollout_timed(w.actions, w.mu0, w.S0, 2, w.include_time, w.time0)
    best = min(eng.rollout_timed(w.actions, w.mu0, w.S0, 5, w.include_time, w.time0)[0] for _ in range(3))
    print(f"{name} B={B}: {best:.3f} ms/launch -> {B / best * 1e3:.0f} rollouts/s", flush=True)
