"""gpmpc_rollout_grad for 8 < D <= 16 (csrc/grad_wide_kernel.h) against the numpy adjoint (oracle/adjoint.py), small sizes.
  python tools/gpu_grad_wide_check.py [time]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gp_mpc_amd
from oracle import synth, adjoint, gpmpc_oracle as orc

eng = gp_mpc_amd.HipEngine(0)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


bad = 0
for (N, D, A, H, B, tm, s0) in [(40, 9, 2, 3, 2, False, 1e-6), (70, 12, 3, 2, 2, False, 1e-5), (128, 16, 4, 3, 2, False, 1e-6),
                                (33, 16, 4, 2, 3, True, 1e-5), (50, 10, 1, 4, 2, False, 1e-3), (17, 16, 4, 2, 2, False, 1e-4)]:
    w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=N + D, s0=s0, time0=2.0 if tm else 0.0)
    f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    out = eng.rollout_grad(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    Jg, gg = out["J"].cpu().numpy(), out["grad"].cpu().numpy()
    eJ = eg = 0.0
    for b in range(B):
        J, g, *_ = adjoint.lcb_and_gradient(f, w.actions[b], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa, w.include_time, w.time0)
        eJ = max(eJ, abs(Jg[b] - J) / abs(J))
        eg = max(eg, rel(gg[b], g))
    ok = eJ < 1e-8 and eg < 1e-6
    bad += not ok
    print(f"N={N} D={D} A={A} H={H} B={B} time={int(tm)} s0={s0:g}: J {eJ:.1e} grad {eg:.1e} {'OK' if ok else 'FAIL'}", flush=True)
    if not ok:
        print("  gpu ", gg[0].ravel()[:8], "\n  ref ", g.ravel()[:8] if B == 1 else adjoint.lcb_and_gradient(f, w.actions[0], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa, w.include_time, w.time0)[1].ravel()[:8])
print("WIDE GRADIENT", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
if "time" in sys.argv[1:]:
    import torch
    for (N, B, H) in [(1024, 4, 2), (4096, 16, 4)]:
        w = synth.make_workload(N, 16, 4, H, B, seed=81)
        eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        eng.set_cost(w.target, w.W, w.W_T, w.kappa)
        eng.rollout_grad(w.actions, w.mu0, w.S0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout_grad(w.actions, w.mu0, w.S0)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        ms, _ = eng.rollout_timed(w.actions, w.mu0, w.S0, 1)
        print(f"N={N} D=16 H={H} B={B}: objective + gradient {tg * 1e3:.1f} ms, forward alone {ms:.1f} ms; moment pass + sweep "
              f"{(tg * 1e3 - ms) / (B * H):.1f} ms per (candidate, step) item with the whole chip", flush=True)
eng.close()
