"""Sequential scipy L-BFGS-B restarts (reference behaviour) vs the batched projected L-BFGS, same starts (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
from scipy.optimize import minimize
from helpers import load, workload_of, make_controller
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
cases = [("golden lcb_grad_norm", workload_of(load("lcb_grad_norm")), 6),
         ("c2 shape", synth.make_workload(200, 3, 1, 25, 2, seed=0), 16)]
for name, w, restarts in cases:
    mu0, S0 = torch.as_tensor(w.mu0), torch.as_tensor(w.S0)
    seq = make_controller(w, optimize=True, restarts=restarts, engine=eng)
    seq._prepare()
    np.random.seed(7)
    H, A = w.actions.shape[1], w.actions.shape[2]
    x0s = [np.random.uniform(0, 1, H * A) for _ in range(restarts)]
    t0 = time.perf_counter()
    res = [minimize(fun=seq.compute_mean_lcb_trajectory, x0=x0, jac=True, args=(mu0, S0), method="L-BFGS-B",
                    bounds=seq.actions_mapper.bounds, options=seq.config.controller.actions_optimizer_params) for x0 in x0s]
    t_seq = time.perf_counter() - t0
    print(f"== {name}: scipy sequential {restarts} restarts: {t_seq*1e3:.1f} ms, nfev {[r.nfev for r in res]}")
    print("   J*:", np.array2string(np.array([r.fun for r in res]), precision=6))
    bat = make_controller(w, optimize=True, restarts=restarts, engine=eng)
    bat.config.controller.candidate_optimizer = "lbfgs"
    bat.config.controller.init_from_previous_actions = False
    it = iter(x0s)
    import importlib
    mod = importlib.import_module(type(bat).__module__)
    orig = mod.generate_mpc_action_init_random
    mod.generate_mpc_action_init_random = lambda len_horizon, dim_action: next(it)
    t0 = time.perf_counter()
    bat._get_optimal_actions(mu0, S0)
    t_bat = time.perf_counter() - t0
    mod.generate_mpc_action_init_random = orig
    print(f"   batched: {t_bat*1e3:.1f} ms, launches {bat.lbfgs_evaluations}")
    print("   J*:", np.array2string(bat.candidates_final_J, precision=6), "identical to sequential:",
          np.array_equal(bat.candidates_final_J, np.array([r.fun for r in res])))
eng.close()
