"""B = 1 forward / objective + gradient / host-in-host-out time per call at c2 and c3 (default dispatch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gp_mpc_amd
from oracle import synth
eng = gp_mpc_amd.HipEngine(0)
for name in ("c2", "c3"):
    n, d, a, h, b, tm = synth.SHAPES[name]
    w = synth.make_workload(n, d, a, h, 2, include_time=tm, seed=0)
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    acts = torch.as_tensor(w.actions[:1], device="cuda:0")
    def t(fn, reps=50):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print(f"{name} B=1 [{os.path.basename(gp_mpc_amd._lib.LIB_PATH)}]: rollout {t(lambda: eng.rollout(acts, w.mu0, w.S0)):.4f} ms (cluster {eng.last_cluster}), "
          f"rollout_grad {t(lambda: eng.rollout_grad(acts, w.mu0, w.S0)):.4f} ms, objective_grad_host {t(lambda: eng.objective_grad_host(w.actions[0], w.mu0, w.S0)):.4f} ms", flush=True)
eng.close()
