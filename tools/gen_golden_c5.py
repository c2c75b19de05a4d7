#!/usr/bin/env python3
"""Config-5 scale fixtures (N=4096, D=16, A=4), computed by the CPU oracle (the reference formulation cannot run
this size: its (D,D,N,N) temporaries are 34 GB each).  The oracle itself is pinned against
reference-generated goldens at small N.  Inputs are regenerated from the seed by oracle/synth.py; only expected
outputs are stored.

  python tools/gen_golden_c5.py            one moment-matched step, 2 candidates  -> oracle_c5_step.npz   (~ minutes)
  python tools/gen_golden_c5.py --steps 5  five horizon steps,     2 candidates  -> oracle_c5_traj.npz   (~ 5x that)
  python tools/gen_golden_c5.py --steps 50 --candidates 1   BASELINE configs[4]'s full horizon, 1 candidate -> oracle_c5_h50.npz
                                                             (round 3; ~ an hour on 8 cores: run it in the background)
  python tools/gen_golden_c5.py --inrange --steps 20 --candidates 4   contracting targets + dense Sigma_0: the state stays inside
                                                             the memory's range over the horizon -> oracle_c5_inrange.npz (round 4)

Reproducibility (checked in a scratch copy, round 4): the one-step fixture regenerates bit for bit; the 20-step in-range fixture to
6e-13 (means) / 5e-12 (covariances) -- the oracle's matrix products run on a threaded BLAS whose summation order depends on the
threads it gets, and twenty moment-matched steps carry that forward.  The GPU tests compare at 1e-9 and looser.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402
from oracle import gpmpc_oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--candidates", type=int, default=2)
ap.add_argument("--inrange", action="store_true")
args = ap.parse_args()
N, D, A, H, B = 4096, 16, 4, args.steps, args.candidates
SEED = 77 if H == 1 else (78 if H <= 5 else 79)
name = "oracle_c5_step.npz" if H == 1 else ("oracle_c5_traj.npz" if H <= 5 else f"oracle_c5_h{H}.npz")
KW = {}
if args.inrange:
    SEED, name, KW = 83, "oracle_c5_inrange.npz", dict(dynamics="contracting", dense_s0=0.02)
w = synth.make_workload(N, D, A, H, B, seed=SEED, **KW)
t0 = time.time()
f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
print("factorised", time.time() - t0, flush=True)
out = orc.evaluate_candidates(f, w)
print("evaluated", time.time() - t0, flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", name),
                    N=N, D=D, A=A, H=H, B=B, seed=SEED, **({"inrange": True} if args.inrange else {}),      # (key set of the round-2/3 fixtures unchanged)
                    beta_head=f.beta[:, :64], mu=out["mu"], Sig=out["Sig"],
                    cost_mu=out["cost_mu"], cost_var=out["cost_var"], J=out["J"],
                    x_checksum=np.array([w.X.sum(), w.Y.sum(), w.actions.sum()]))
print("J", out["J"])
