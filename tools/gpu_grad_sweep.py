"""A/B sweep of the gradient's LDS-resident moment pass: gpmpc_rollout_grad at BASELINE shapes under the options
grad_chunk_rows (0 = the schedule model's choice) x grad_share_cu (2 never / 1 two workgroups per CU).  Prints one line per
combination (median of `reps` launches, HIP events around the launch) and the relative error against the default's gradient."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import gp_mpc_amd
from oracle import synth

shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c1", "c3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
chunks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 64, 56, 48, 44, 40, 32, 24, 16, 8]
for shape in shapes:
    name, _, bs = shape.partition(":")
    n, d, a, h, b, tm = synth.SHAPES[name]
    B = int(bs) if bs else b
    w = synth.make_workload(n, d, a, h, B, include_time=tm, seed=0)
    eng = gp_mpc_amd.HipEngine(0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    ref = None
    for share in (2, 1):
        for rows in chunks:
            if rows > ((n + 3) & ~3):
                continue
            eng.set_option("grad_share_cu", share)
            eng.set_option("grad_chunk_rows", rows)
            out = eng.rollout_grad(acts, w.mu0, w.S0, w.include_time, w.time0)
            torch.cuda.synchronize()
            gr = out["grad"].cpu().numpy()
            if ref is None:
                ref = gr
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.rollout_grad(acts, w.mu0, w.S0, w.include_time, w.time0)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            err = float(np.max(np.abs(gr - ref)) / np.max(np.abs(ref)))
            print(f"{name} N={n} B={B} share_cu={share} chunk_rows={rows}: rollout_grad {np.median(ts):.3f} ms (min {min(ts):.3f}), grad path {eng.last_grad_path}, vs default {err:.1e}", flush=True)
    eng.close()
