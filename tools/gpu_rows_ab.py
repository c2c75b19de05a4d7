"""Rows per chunk of the pairwise items (option rows_per_chunk), interleaved best-of-3 (HIP-event kernel time)."""
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
    eng.rollout_timed(acts, w.mu0, w.S0, 40 if n < 600 else 5, w.include_time, w.time0)
    best = {}
    for rep in range(3):
        for rows in (0, 8, 12, 16, 24, 32, 48, 64):
            eng.set_option("rows_per_chunk", rows)
            ms, J = eng.rollout_timed(acts, w.mu0, w.S0, 15 if n < 600 else 3, w.include_time, w.time0)
            best[rows] = min(best.get(rows, 1e9), ms)
    eng.set_option("rows_per_chunk", 0)
    print(f"{shape} N={n} D={d} B={B}: " + "  ".join(f"rows={k or 'auto'}: {v:.3f}" for k, v in best.items()), flush=True)
eng.close()
