"""A/B of the compile-time-D vs runtime-D instantiation of the rollout kernel (HIP-event kernel time)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for (N, D, A, H, B) in ((200, 3, 1, 25, 256), (200, 2, 1, 25, 256), (1000, 4, 2, 30, 256), (300, 6, 2, 10, 256), (300, 8, 3, 10, 256), (200, 16, 4, 5, 256)):
    w = synth.make_workload(N, D, A, H, B, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    line = f"N={N} D={D} H={H} B={B}:"
    for mode in (1, 2):
        eng.set_option("exact_dim", mode)
        eng.rollout_timed(acts, w.mu0, w.S0, 1)
        ms, J = eng.rollout_timed(acts, w.mu0, w.S0, 3)
        line += f"  {'compile-time D' if mode == 1 else 'runtime D'}: {ms:.3f} ms"
    print(line, flush=True)
eng.close()
