#!/usr/bin/env python3
"""Config-5 scale fixture (N=4096, D=16, A=4): one moment-matched step for 2 candidates, computed by the
CPU oracle (the reference formulation cannot run this size: its (D,D,N,N) temporaries are 34 GB each,
SURVEY F7).  The oracle itself is pinned against reference-generated goldens at small N.
Inputs are regenerated from the seed by oracle/synth.py; only expected outputs are stored."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth
from oracle import gpmpc_oracle as orc

N, D, A, H, B, SEED = 4096, 16, 4, 1, 2, 77
w = synth.make_workload(N, D, A, H, B, seed=SEED)
t0 = time.time()
f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
print("factorised", time.time() - t0, flush=True)
out = orc.evaluate_candidates(f, w)
print("evaluated", time.time() - t0, flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_c5_step.npz"),
                    N=N, D=D, A=A, H=H, B=B, seed=SEED, beta_head=f.beta[:, :64], mu=out["mu"], Sig=out["Sig"],
                    cost_mu=out["cost_mu"], cost_var=out["cost_var"], J=out["J"],
                    x_checksum=np.array([w.X.sum(), w.Y.sum(), w.actions.sum()]))
print("J", out["J"])
