"""TEST INFRASTRUCTURE ONLY -- seeded synthetic GP-MPC workloads.

Values live in the normalised [0, 1] ranges the reference works in
(reference: examples/pendulum/config_pendulum.py:12-45 for the hyper-parameter
magnitudes, rl_gp_mpc/control_objects/actions_mappers/action_init_functions.py:4-5
for how candidate action sequences are drawn).
"""
from dataclasses import dataclass, field

import numpy as np

# name -> (N, D, A, H, B, include_time)  -- BASELINE.json configs[0..4]
SHAPES = {
    "c1": (50, 3, 1, 15, 1, False),
    "c2": (200, 3, 1, 25, 256, False),
    "c3": (500, 2, 1, 40, 1024, False),
    "c4": (1000, 4, 2, 30, 2048, False),
    "c5": (4096, 16, 4, 50, 8192, False),
}


@dataclass
class Workload:
    X: np.ndarray            # (N, E) GP inputs   [state, action, (time)]
    Y: np.ndarray            # (N, D) GP targets  (state change)
    lengthscales: np.ndarray  # (D, E)
    outputscales: np.ndarray  # (D,)
    noises: np.ndarray        # (D,)  likelihood noise VARIANCE
    actions: np.ndarray       # (B, H, A) in [0, 1]
    mu0: np.ndarray           # (D,)
    S0: np.ndarray            # (D, D)
    include_time: bool = False
    time0: float = 0.0
    # quadratic cost (reference reward_config.py:58-64)
    target: np.ndarray = field(default=None)      # (D + A,)
    W: np.ndarray = field(default=None)           # (D + A, D + A)
    W_T: np.ndarray = field(default=None)         # (D, D)
    kappa: float = 1.0

    @property
    def dims(self):
        N, E = self.X.shape
        D = self.Y.shape[1]
        B, H, A = self.actions.shape
        return N, D, A, E, H, B


def _rep(vals, n):
    vals = list(vals)
    return np.array([vals[i % len(vals)] for i in range(n)], dtype=np.float64)


def make_workload(N, D, A, H, B, include_time=False, seed=0, noise_var=1e-5,
                  outputscale=5e-2, s0=1e-6, time0=0.0, dynamics="drift", dense_s0=0.0):
    """`dynamics`: "drift" = the SURVEY 8(d) targets (a smooth state change of up to 0.05 per step: over a long horizon the
    mean leaves the [0, 1] range the memory covers and the GP falls back to its prior); "contracting" = targets that pull
    every state towards 0.5 (y_d = -0.2 (x_d - 0.5) + 0.02 sin(...)), so a long rollout stays where the memory is and the
    late steps exercise the data-dependent terms.  `dense_s0` > 0: a dense initial covariance G G^T + s0 I, G ~ dense_s0 N(0, 1)."""
    E = D + A + (1 if include_time else 0)
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.0, 1.0, size=(N, E))
    if include_time:
        X[:, -1] = np.arange(N, dtype=np.float64)      # control-iteration index
    Y = np.empty((N, D))
    for d in range(D):
        if dynamics == "contracting":
            Y[:, d] = -0.2 * (X[:, d] - 0.5) + 0.02 * np.sin(3.0 * X[:, d] + X[:, D + A - 1]) + 1e-3 * rng.standard_normal(N)
        else:
            Y[:, d] = 0.05 * np.sin(3.0 * X[:, d] + X[:, D + A - 1]) + 1e-3 * rng.standard_normal(N)
    ls = 0.5 + rng.uniform(0.0, 1.0, size=(D, E))
    if include_time:
        ls[:, -1] = 100.0 + 50.0 * rng.uniform(0.0, 1.0, size=D)
    rng_a = np.random.default_rng(seed + 1)
    actions = rng_a.uniform(0.0, 1.0, size=(B, H, A))
    mu0 = rng.uniform(0.0, 1.0, size=D)
    S0 = s0 * np.eye(D)
    if dense_s0 > 0.0:
        G = dense_s0 * np.random.default_rng(seed + 2).standard_normal((D, D))
        S0 = G @ G.T + s0 * np.eye(D)
    # cost weights: pendulum example extended by repetition
    target = np.concatenate([_rep([1.0, 0.5, 0.5], D), _rep([0.5], A)])
    W = np.diag(np.concatenate([_rep([1.0, 0.1, 0.1], D), _rep([1e-3], A)]))
    W_T = np.diag(_rep([5.0, 2.0, 2.0], D))
    return Workload(X=X, Y=Y, lengthscales=ls, outputscales=np.full(D, outputscale),
                    noises=np.full(D, noise_var), actions=actions, mu0=mu0, S0=S0,
                    include_time=include_time, time0=float(time0 if include_time else 0.0),
                    target=target, W=W, W_T=W_T, kappa=1.0)


def named(name, seed=0, N=None, B=None, H=None):
    n, d, a, h, b, t = SHAPES[name]
    return make_workload(N or n, d, a, H or h, B or b, include_time=t, seed=seed)
