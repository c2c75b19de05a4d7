"""Cooperative form at B = 1: time per launch over (threads per workgroup, rows per chunk, cluster size)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
shapes = {"c2": (200, 3, 1, 25), "c3": (500, 2, 1, 40), "c1": (50, 3, 1, 15), "n100": (100, 3, 1, 15), "n350d4": (350, 4, 2, 15)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c3", "c1"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name in names:
    n, d, a, h = shapes[name]
    w = synth.make_workload(n, d, a, h, max(B, 2), seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions[:B], device="cuda:0")

    def t_ms():
        eng.rollout(acts, w.mu0, w.S0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            eng.rollout(acts, w.mu0, w.S0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3
    eng.set_option("threads", 0); eng.set_option("rows_per_chunk", 0); eng.set_option("cluster", 1)
    print(f"{name} B={B}: plain default {t_ms():.3f} ms", flush=True)
    for nt in (1024, 512):
        eng.set_option("threads", nt)
        for rpc in ((8, 16, 32) if B == 1 else (16, 32)):
            eng.set_option("rows_per_chunk", rpc)
            line = f"{name} threads {nt} rows/chunk {rpc}:"
            for cs in ((1, 4, 8, 16, 32) if B == 1 else (1, 2, 4, 8, 16)):
                eng.set_option("cluster", cs)
                t = t_ms()
                line += f"  cs{eng.last_cluster} {t:.3f}"
            print(line, flush=True)
eng.close()
