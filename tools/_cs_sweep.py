import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
shapes = {"c1": (50, 3, 1, 15, False), "n80d3": (80, 3, 1, 15, False), "n100d3": (100, 3, 1, 15, False), "n120d3": (120, 3, 1, 15, False), "n130d3": (130, 3, 2, 15, False), "n100d2": (100, 2, 1, 15, False), "n150d2": (150, 2, 1, 15, False), "n200d2": (200, 2, 1, 15, False), "n200d1": (200, 1, 1, 15, False), "n400d1": (400, 1, 1, 15, False), "n100d4": (100, 4, 2, 15, False), "n200d4": (200, 4, 2, 15, False)}
def t(fn, reps=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, (n, d, a, h, tm) in shapes.items():
    w = synth.make_workload(n, d, a, h, 4, include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    for B in (1,):
        acts = torch.as_tensor(w.actions[:B], device="cuda:0")
        line = f"{name} B={B}:"
        for cs in (0, 1, 4, 6, 9, 12, 14, 16, 24):
            eng.set_option("cluster", cs)
            ms = t(lambda: eng.rollout(acts, w.mu0, w.S0, w.include_time, w.time0))
            line += f"  cs{cs}->{eng.last_cluster} {ms:.3f}"
        print(line, flush=True)
eng.close()
