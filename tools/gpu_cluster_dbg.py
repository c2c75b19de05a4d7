"""Timing experiments of the cooperative form's exchange (debug library built with -DGPMPC_CL_DEBUG):
cluster_debug 0 = real exchange, 1 = no wait at all, 2 = one read per value, no tag check (results are garbage for 1 / 2)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for name, n, d, a, h in (("c2", 200, 3, 1, 25), ("c3", 500, 2, 1, 40), ("c1", 50, 3, 1, 15)):
    w = synth.make_workload(n, d, a, h, 4, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions[:1], device="cuda:0")
    for rpc in (8, 16, 32):
        eng.set_option("rows_per_chunk", rpc)
        for cs in (1, 4, 8, 16):
            eng.set_option("cluster", cs)
            line = f"{name} rows/chunk {rpc} cluster {cs}:"
            for dbg in ((0,) if cs == 1 else (0, 2, 1)):
                eng.set_option("cluster_debug", dbg)
                eng.rollout(acts, w.mu0, w.S0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    eng.rollout(acts, w.mu0, w.S0)
                torch.cuda.synchronize()
                line += f"  dbg{dbg} {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms"
            print(line, flush=True)
eng.close()
