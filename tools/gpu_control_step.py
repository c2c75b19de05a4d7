"""End-to-end time of one control step's optimisation (reference default: optimize=True, scipy L-BFGS-B, jac=True) on a
config-2-sized memory: seconds per get_optimal_actions, per objective+gradient evaluation, and the pure launch time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
from helpers import make_controller
import gp_mpc_amd
from oracle import synth

eng = gp_mpc_amd.HipEngine(0)
w = synth.make_workload(200, 3, 1, 25, 2, seed=0)
mu0, S0 = torch.as_tensor(w.mu0), torch.as_tensor(w.S0)
# (the single-call figures first: after the 16 solver threads of the lockstep runs below the interpreter times them several times too long)
c = make_controller(w, optimize=True, restarts=1, engine=eng)
c._prepare()
c.transition_model.set_cost(c.config.reward)
acts = torch.as_tensor(w.actions[:1], device="cuda:0")
eng.rollout_grad(acts, w.mu0, w.S0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.rollout_grad(acts, w.mu0, w.S0)
torch.cuda.synchronize()
print(f"engine.rollout_grad B=1 (device-resident actions): {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
x = w.actions[0].reshape(-1)
c.compute_mean_lcb_trajectory(x, mu0, S0)
t0 = time.perf_counter()
for _ in range(20):
    c.compute_mean_lcb_trajectory(x, mu0, S0)
print(f"controller.compute_mean_lcb_trajectory: {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
for restarts, opt in ((1, None), (4, None), (4, "lbfgs"), (16, "lbfgs")):
    c = make_controller(w, optimize=True, restarts=restarts, engine=eng)
    c.config.controller.candidate_optimizer = opt
    np.random.seed(1)
    c._get_optimal_actions(mu0, S0)                 # warm-up (allocations, first launches)
    # (the lockstep solves hand the interpreter lock from thread to thread once per evaluation round: their wall time moves with the
    #  host's thread scheduling -- three timed repetitions, all printed)
    times = []
    for rep in range(3 if opt else 1):
        n0 = c.num_rollouts
        np.random.seed(2)
        t0 = time.perf_counter()
        c._get_optimal_actions(mu0, S0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        nev = c.num_rollouts - n0
    dt = min(times)
    print(f"restarts={restarts} optimizer={opt or 'scipy sequential'}: {dt*1e3:.1f} ms per control-step optimisation, {nev} evaluations, "
          f"{dt/nev*1e3:.3f} ms per evaluation" + (f", {c.lbfgs_evaluations} launches; repetitions " + " / ".join(f"{t*1e3:.1f}" for t in times) + " ms" if opt else ""), flush=True)
eng.close()
