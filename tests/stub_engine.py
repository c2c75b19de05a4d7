"""CPU stand-in for HipEngine in the multi-process (gloo) tests: same method set as far as the controller and
sharding.py use it, tensors on the CPU, arithmetic by the oracle.  TEST CODE ONLY -- the product has no CPU engine;
what these tests exercise is everything AROUND the launches: slice arithmetic, empty slices, the packed records, the
single all_gather, the cross-rank keep-the-best rule, the logging caches on every rank."""
import math

import numpy as np
import torch

from oracle import gpmpc_oracle as orc


class OracleEngine:
    def __init__(self):
        self.device = torch.device("cpu")
        self._cost = None
        self.launches = 0

    def prepare(self, X, Y, lengthscales, outputscales, noises):
        n = lambda a: np.asarray(a, dtype=np.float64)                          # noqa: E731
        self.f = orc.Factors(n(X), n(Y), n(lengthscales), n(outputscales).reshape(-1), n(noises).reshape(-1))
        self.D = self.f.Y.shape[1]

    def set_cost(self, target, W, W_T, kappa, clip_to_zero=False, state_min=None, state_max=None):
        self._cost = (np.asarray(target), np.asarray(W), np.asarray(W_T), float(kappa), bool(clip_to_zero), state_min, state_max)

    def rollout(self, actions, mu0, S0, include_time=False, time0=0.0, trajectories=True, stage_costs=True, out=None):
        actions = np.asarray(actions, dtype=np.float64)
        assert actions.shape[0] >= 1, "the real engine rejects B = 0"
        self.launches += 1
        mu, Sig = orc.predict_trajectory(self.f, actions, np.asarray(mu0), np.asarray(S0), include_time, time0)
        target, W, W_T, kappa, clip, smin, smax = self._cost
        cm, cv = orc.stage_costs(mu, Sig, actions, target, W, W_T, smin, smax)
        J = orc.lcb_objective(cm, cv, kappa, clip)
        t = torch.as_tensor
        return {"J": t(J), "mu": t(mu), "Sig": t(Sig), "cost_mu": t(cm), "cost_var": t(cv)}

    def argmin_async(self, J, first_global_index=0, actions=None, out=None):
        """Record [best J, global index (-1: none), winning sequence] by the rule of gpmpc_argmin_async."""
        J = np.asarray(J, dtype=np.float64)
        best, val = -1, math.inf
        for k, v in enumerate(J):
            if first_global_index + k == 0 and np.isnan(v):
                best, val = 0, v
                break
            if v < val:
                best, val = first_global_index + k, v
        ha = 0 if actions is None else int(actions[0].numel())
        rec = torch.zeros(2 + ha, dtype=torch.float64) if out is None else out
        rec[0], rec[1] = val, float(best)
        if ha and best >= 0:
            rec[2:] = actions[best - first_global_index].reshape(-1)
        return rec
