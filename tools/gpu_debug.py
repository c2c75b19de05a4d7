import sys, time, faulthandler
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
faulthandler.dump_traceback_later(25, exit=True)
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:7.2f}s]", *a, flush=True)
import numpy as np, torch
log("torch imported", torch.cuda.is_available())
import gp_mpc_amd
from helpers import load, workload_of, factors_of, rel_err
eng = gp_mpc_amd.HipEngine(0)
log("engine created")
g = load("traj_c1"); w = workload_of(g); f = factors_of(w)
eng.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
log("set_factors done")
eng.set_cost(w.target, w.W, w.W_T, w.kappa)
log("set_cost done")
stage = sys.argv[1] if len(sys.argv) > 1 else "rollout"
if stage == "rollout":
    out = eng.rollout(w.actions[:1, :1], w.mu0, w.S0)
    torch.cuda.synchronize()
    log("rollout B=1 H=1 done", out["J"].cpu().numpy(), out["mu"].cpu().numpy()[0, 1], g["mu"][0, 1])
    out = eng.rollout(w.actions, w.mu0, w.S0)
    torch.cuda.synchronize()
    log("rollout full done", rel_err(out["mu"].cpu().numpy(), g["mu"]), rel_err(out["Sig"].cpu().numpy(), g["Sig"]), rel_err(out["J"].cpu().numpy(), g["J"]))
else:
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    log("prepare done")
    iK, beta = eng.factors()
    log("factors", rel_err(iK.cpu().numpy(), g["iK"]), rel_err(beta.cpu().numpy(), g["beta"]))
