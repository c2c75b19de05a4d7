"""Accuracy and time of the border-updated factors vs a fresh factorisation (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import gp_mpc_amd
from oracle import synth, gpmpc_oracle as orc
from helpers import rel_err

for (N0, D, A, tm, nsteps) in [(40, 3, 1, False, 16), (200, 3, 1, False, 32), (130, 4, 2, True, 32), (500, 6, 2, False, 32)]:
    w = synth.make_workload(N0 + nsteps, D, A, 4, 4, include_time=tm, seed=N0)
    inc, full = gp_mpc_amd.HipEngine(0), gp_mpc_amd.HipEngine(0)
    full.set_option("incremental", 0)
    inc.set_option("refresh_every", 1000)
    dev = torch.device("cuda", 0)
    X, Y = torch.as_tensor(w.X, device=dev), torch.as_tensor(w.Y, device=dev)
    inc.prepare(X[:N0], Y[:N0], w.lengthscales, w.outputscales, w.noises)
    for n in range(N0 + 1, N0 + nsteps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        inc.prepare(X[:n], Y[:n], w.lengthscales, w.outputscales, w.noises)
        torch.cuda.synchronize(); ti = time.perf_counter() - t0
        assert inc.last_prepare_mode == 1
        if (n - N0) in (1, 2, 4, 8, 16, 32):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            full.prepare(X[:n], Y[:n], w.lengthscales, w.outputscales, w.noises)
            torch.cuda.synchronize(); tf = time.perf_counter() - t0
            iK, beta = inc.factors(); iK0, beta0 = full.factors()
            iKo, betao = orc.factorize(w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)
            K = orc.rbf_ard_gram(w.X[:n], w.lengthscales, w.outputscales) + np.asarray(w.noises)[:, None, None] * np.eye(n)
            cond = max(np.linalg.cond(K[a]) for a in range(D))
            print(f"N0={N0} D={D} +{n-N0}: inc {ti*1e3:.3f} ms full {tf*1e3:.3f} ms | iK inc-full {rel_err(iK.cpu().numpy(), iK0.cpu().numpy()):.2e} "
                  f"inc-orc {rel_err(iK.cpu().numpy(), iKo):.2e} full-orc {rel_err(iK0.cpu().numpy(), iKo):.2e} | beta inc-full "
                  f"{rel_err(beta.cpu().numpy(), beta0.cpu().numpy()):.2e} inc-orc {rel_err(beta.cpu().numpy(), betao):.2e} "
                  f"full-orc {rel_err(beta0.cpu().numpy(), betao):.2e} cond {cond:.1e}", flush=True)
    inc.close(); full.close()
