"""Config 5 (N = 4096, D = 16, A = 4, B = 256): ONE horizon step of the streaming kernel started from the state the full-size
oracle fixture (tests/golden/oracle_c5_h50.npz, seed 79) holds at horizon step t -- early states (Sigma ~ 1e-6 I) against
late ones (the state has left the memory's range, Sigma diag ~ 1) -- to reconcile the per-step time of an H = 1 launch
(profiles/r03_c5_N4096_kernel_trace_stats.txt: 0.495 s) with the bench's 27.9 s / 50 steps = 0.558 s.
  python tools/gpu_c5_late.py [t ...] [name=value engine options]      default t: 0 2 5 10 25 45"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gp_mpc_amd
from oracle import synth

steps = [int(a) for a in sys.argv[1:] if "=" not in a] or [0, 2, 5, 10, 25, 45]
fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "oracle_c5_h50.npz")))
N, D, A, B = 4096, 16, 4, 256
w = synth.make_workload(N, D, A, 50, B, seed=79)
assert np.allclose([w.X.sum(), w.Y.sum(), w.actions[:1].sum()], fx["x_checksum"], rtol=0, atol=1e-9)
eng = gp_mpc_amd.HipEngine(0)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
eng.set_cost(w.target, w.W, w.W_T, w.kappa)
for t in steps:
    mu, Sig = fx["mu"][0, t], fx["Sig"][0, t]
    acts = np.ascontiguousarray(w.actions[:, t:t + 1])
    ms, J = eng.rollout_timed(acts, mu, Sig, 1)
    print(f"state of horizon step {t:2d}: Sigma diag {np.diag(Sig).min():.2e} .. {np.diag(Sig).max():.2e}, |mu - 0.5| max {np.abs(mu - 0.5).max():.2f}: "
          f"{ms:.1f} ms per horizon step of {B} candidates", flush=True)
