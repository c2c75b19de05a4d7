#!/usr/bin/env python3
"""Config-5 scale gradient fixture (N = 4096, D = 16, A = 4, H = 2, one candidate): J and dJ/du from the numpy adjoint
(oracle/adjoint.py, pinned against the reference's autograd goldens at small sizes and against torch autograd at
D = 9 and D = 16: tests/test_oracle_vs_golden.py).  The reference itself cannot run this size.
Inputs are regenerated from the seed by oracle/synth.py; only expected outputs are stored.

  python tools/gen_golden_c5_grad.py      -> tests/golden/oracle_c5_grad.npz   (~ 10-20 minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth, adjoint  # noqa: E402
from oracle import gpmpc_oracle as orc  # noqa: E402

N, D, A, H, B, SEED = 4096, 16, 4, 2, 1, 81
w = synth.make_workload(N, D, A, H, B, seed=SEED)
t0 = time.time()
f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
print("factorised", time.time() - t0, flush=True)
J, grad, mus, Sigs, _ = adjoint.lcb_and_gradient(f, w.actions[0], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa)
print("adjoint", time.time() - t0, flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_c5_grad.npz"), N=N, D=D, A=A, H=H, B=B, seed=SEED,
                    J=J, grad=grad, mu=mus, Sig=Sigs, x_checksum=np.array([w.X.sum(), w.Y.sum(), w.actions.sum()]))
print("J", J, "grad", grad)
