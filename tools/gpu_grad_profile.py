"""Workload for profiling the gradient path: gpmpc_rollout_grad at a BASELINE shape (default config 2)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
opts = dict(kv.split("=") for kv in sys.argv[4:])           # engine options: name=value ...
n, d, a, h, b, tm = synth.SHAPES[shape]
w = synth.make_workload(n, d, a, h, B, include_time=tm, seed=0)
eng = gp_mpc_amd.HipEngine(0)
eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
eng.set_cost(w.target, w.W, w.W_T, w.kappa)
for k, v in opts.items():
    eng.set_option(k, int(v))
acts = torch.as_tensor(w.actions, device="cuda:0")
eng.rollout_grad(acts, w.mu0, w.S0, w.include_time, w.time0)
torch.cuda.synchronize()
for name, fn in (("rollout_grad", lambda: eng.rollout_grad(acts, w.mu0, w.S0, w.include_time, w.time0)),
                 ("rollout", lambda: eng.rollout(acts, w.mu0, w.S0, w.include_time, w.time0))):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{shape} B={B} {opts}: {name} {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per launch", flush=True)
eng.close()
