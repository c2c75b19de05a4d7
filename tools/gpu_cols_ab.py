"""A/B of one vs two columns per lane in the pairwise pass of the rollout kernel (HIP-event kernel time)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
for shape, B in (("c1", 256), ("c2", 256), ("c2", 1024), ("c3", 512), ("c4", 256)):
    n, d, a, h, b, tm = synth.SHAPES[shape]
    w = synth.make_workload(n, d, a, h, B, include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    line = f"{shape} N={n} D={d} H={h} B={B}:"
    ref = None
    for cols, rows in ((1, 0), (2, 0), (2, 32), (2, 24), (2, 16)):
        eng.set_option("cols_per_lane", cols)
        eng.set_option("rows_per_chunk", rows)
        ms, J = eng.rollout_timed(acts, w.mu0, w.S0, 5 if n < 600 else 2, w.include_time, w.time0)
        if ref is None:
            ref = J
        err = float(((J - ref).abs() / ref.abs()).max())
        line += f"  cols={cols} rows={rows or 'auto'}: {ms:.3f} ms ({B / ms:.0f} k/s, dJ {err:.1e})"
    print(line, flush=True)
    eng.set_option("rows_per_chunk", 0)
eng.close()
