import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for (N, D, A, H, B) in [(1024, 16, 4, 1, 256), (2048, 16, 4, 1, 256), (4096, 16, 4, 1, 256), (4096, 16, 4, 2, 256)]:
    w = synth.make_workload(N, D, A, H, B, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 1)
    print(f"N={N} D={D} H={H} B={B}: {ms:.1f} ms/launch = {ms/H:.1f} ms per horizon step of 256 candidates; J[0]={float(J[0]):.6g}", flush=True)
