import sys
sys.path.insert(0, '.')
import numpy as np, torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name, B in [("c2", 256), ("c3", 256), ("c4", 256)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    for rpc in (0, 16, 24, 32, 48, 64):
        eng.set_option("rows_per_chunk", rpc)
        ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 2)
        best = min(eng.rollout_timed(w.actions, w.mu0, w.S0, 5)[0] for _ in range(3))
        print(f"{name} B={B} rows_per_chunk={rpc}: {best:.3f} ms/launch", flush=True)
    eng.set_option("rows_per_chunk", 0)
# larger batches: two workgroups per CU hide each other's serial phases
for name, B in [("c2", 512), ("c2", 1024), ("c2", 2048), ("c3", 1024), ("c4", 2048)]:
    w = synth.named(name, B=B)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    for thr in (0, 512, 1024):
        eng.set_option("threads", thr)
        ms, J = eng.rollout_timed(w.actions, w.mu0, w.S0, 1)
        best = min(eng.rollout_timed(w.actions, w.mu0, w.S0, 3)[0] for _ in range(2))
        print(f"{name} B={B} threads={thr}: {best:.3f} ms/launch -> {B/best*1e3:.0f} rollouts/s", flush=True)
    eng.set_option("threads", 0)
