"""Shared test helpers (fixtures -> oracle objects, error metrics)."""
import os

import numpy as np

from oracle import synth
from oracle.gpmpc_oracle import Factors

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def workload_of(g):
    return synth.Workload(X=g["X"], Y=g["Y"], lengthscales=g["lengthscales"], outputscales=g["outputscales"],
                          noises=g["noises"], actions=g["actions"], mu0=g["mu0"], S0=g["S0"],
                          include_time=bool(g["include_time"]), time0=float(g["time0"]), target=g["target"],
                          W=g["W"], W_T=g["W_T"], kappa=float(g["kappa"]))


def factors_of(w):
    return Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)


def rel_err(a, b):
    """max |a-b| / max|b|  (scale-relative, robust to entries that pass through zero)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))
