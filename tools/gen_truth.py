#!/usr/bin/env python3
"""Extended-precision values for the reference-made trajectory goldens (VERDICT r1 "weak" 1).

For every named golden in tests/golden/ this evaluates the same hot path on the same fp64 inputs with every
operation in longdouble (oracle/extended_precision.py; unit round-off 5.4e-20) and writes
tests/golden/<name>_truth.npz = {mu, Sig} rounded to fp64, plus the distances
    ref_err_mu / ref_err_Sig      |reference golden - extended| / max|extended|
    oracle_err_mu / oracle_err_Sig |numpy fp64 oracle - extended| / max|extended|
so that tests can require |HIP - extended| <= |reference - extended| (and print both) instead of trusting a
tolerance on |HIP - reference| alone.

  python tools/gen_truth.py [names...]      default: traj_c3 traj_c2 traj_c4 traj_c4_n1000
  python tools/gen_truth.py --self-check    how good is the extended value itself?  (a) memory points permuted
                                            (another summation order, another pivot order), (b) one step at
                                            N = 40 against 50-digit mpmath
Needs only the committed fixtures and oracle/ (not /root/reference).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import extended_precision as xp  # noqa: E402
from oracle import gpmpc_oracle as orc  # noqa: E402
from oracle import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a = np.asarray(a, dtype=np.longdouble)
    b = np.asarray(b, dtype=np.longdouble)
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def extended_trajectories(g, perm=None):
    X, Y = g["X"], g["Y"]
    if perm is not None:
        X, Y = X[perm], Y[perm]
    f = xp.Factors(X, Y, g["lengthscales"], g["outputscales"], g["noises"])
    mus, Sigs = [], []
    for b in range(g["actions"].shape[0]):
        mu, Sig = xp.predict_trajectory(f, g["actions"][b], g["mu0"], g["S0"], bool(g["include_time"]), float(g["time0"]))
        mus.append(mu)
        Sigs.append(Sig)
    return np.stack(mus), np.stack(Sigs)


def one(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    t0 = time.time()
    mu, Sig = extended_trajectories(g)
    fo = orc.Factors(g["X"], g["Y"], g["lengthscales"], g["outputscales"], g["noises"])
    omu, oSig = orc.predict_trajectory(fo, g["actions"], g["mu0"], g["S0"], bool(g["include_time"]), float(g["time0"]))
    d = dict(mu=mu.astype(np.float64), Sig=Sig.astype(np.float64),
             ref_err_mu=rel(g["mu"], mu), ref_err_Sig=rel(g["Sig"], Sig),
             oracle_err_mu=rel(omu, mu), oracle_err_Sig=rel(oSig, Sig),
             ref_vs_oracle_Sig=rel(oSig, g["Sig"]))
    np.savez_compressed(os.path.join(GOLDEN, name + "_truth.npz"), **d)
    print(f"{name}: {time.time() - t0:.0f} s  |ref - ext| mu {d['ref_err_mu']:.2e} Sig {d['ref_err_Sig']:.2e}   "
          f"|oracle - ext| mu {d['oracle_err_mu']:.2e} Sig {d['oracle_err_Sig']:.2e}   |ref - oracle| Sig {d['ref_vs_oracle_Sig']:.2e}",
          flush=True)


def self_check():
    g = dict(np.load(os.path.join(GOLDEN, "traj_c3.npz")))
    g["actions"] = g["actions"][:1]
    mu, Sig = extended_trajectories(g)
    perm = np.random.default_rng(0).permutation(g["X"].shape[0])
    mu2, Sig2 = extended_trajectories(g, perm)
    print(f"traj_c3 (N=500, H=40), memory points permuted: extended vs extended  mu {rel(mu2, mu):.2e}  Sig {rel(Sig2, Sig):.2e}")
    w = synth.make_workload(40, 2, 1, 1, 1, seed=5, s0=1e-4)
    E = w.X.shape[1]
    m = np.concatenate([w.mu0, w.actions[0, 0]])
    s = np.zeros((E, E))
    s[:2, :2] = [[2e-4, 5e-5], [5e-5, 1e-4]]
    Mm, Sm = xp.mp_single_step(w.X, w.Y, w.lengthscales, w.outputscales, w.noises, m, s)
    f = xp.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    M, S, _ = xp.moment_match_step(f, xp._ld(m), xp._ld(s))
    fo = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    Mo, So, _ = orc.moment_match_step(fo, m[None], s[None])
    print(f"one step, N=40, vs 50-digit mpmath: extended M {rel(M, Mm):.2e} S {rel(S, Sm):.2e};  "
          f"fp64 oracle M {rel(Mo[0], Mm):.2e} S {rel(So[0], Sm):.2e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["traj_c3", "traj_c2", "traj_c4", "traj_c4_n1000"])
    ap.add_argument("--self-check", action="store_true")
    a = ap.parse_args()
    if a.self_check:
        self_check()
    else:
        for n in a.names:
            one(n)
