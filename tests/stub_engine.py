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

    # -- the two halves of a sharded cross-entropy iteration, restating csrc/search.hip (cem_sample / cem_map / cem_elites /
    #    cem_merge kernels) in numpy; draws must be supplied (`noise`), the Philox generator exists on the device only
    def cem_local(self, mu0, S0, B_total, first, B_local, H, A, iteration, n_elite, state, seed=0, include_time=False,
                  time0=0.0, first_candidate=None, max_change=None, action_prev=None, noise=None, out=None):
        assert noise is not None, "the CPU stand-in has no Philox generator"
        n = H * A
        st = state.numpy()
        mean, std, best = st[:n], st[n:2 * n], st[2 * n:3 * n]
        rec = np.zeros((n_elite, n + 2))
        rec[:, 0], rec[:, 1] = np.inf, 2147483647.0
        if B_local > 0:
            draw = np.asarray(noise, dtype=np.float64)[iteration, first:first + B_local]
            X = draw.copy() if iteration == 0 else np.clip(mean + std * draw, 0.0, 1.0)
            if first == 0:
                if iteration > 0:
                    X[0] = best
                elif first_candidate is not None:
                    X[0] = first_candidate
            if max_change is None:
                acts = X.reshape(B_local, H, A)
            else:
                m = np.asarray(max_change)
                acts = np.clip(np.asarray(action_prev) + np.cumsum(X.reshape(B_local, H, A) * 2.0 * m - m, axis=1), 0.0, 1.0)
            J = self.rollout(acts, mu0, S0, include_time, time0)["J"].numpy()
            J = np.where(np.isnan(J), np.inf, J)
            order = np.lexsort((np.arange(B_local), J))[:n_elite]
            k = len(order)
            rec[:k, 0], rec[:k, 1], rec[:k, 2:] = J[order], order + first, X[order]
        t = torch.as_tensor(rec)
        if out is not None:
            out.copy_(t)
            return out
        return t

    def cem_merge(self, elites, n_elite, n, iteration, state):
        rec = elites.numpy().reshape(-1, n + 2)
        order = np.lexsort((rec[:, 1], rec[:, 0]))[:n_elite]
        st = state.numpy()
        top = rec[order, 2:]
        s = np.zeros(n)
        for e in range(n_elite):                  # the kernel's summation order
            s = s + top[e]
        m = s / n_elite
        q = np.zeros(n)
        for e in range(n_elite):
            q = q + (top[e] - m) ** 2
        if iteration == 0 or rec[order[0], 0] < st[3 * n]:
            st[2 * n:3 * n] = top[0]
            st[3 * n] = rec[order[0], 0]
        st[:n] = m
        st[n:2 * n] = np.sqrt(q / n_elite) + 1e-3

    # -- objective + gradient (oracle/adjoint.py) in the packed form HipEngine.rollout_grad returns
    def rollout_grad(self, actions, mu0, S0, include_time=False, time0=0.0, trajectories=False):
        from oracle import adjoint
        actions = np.asarray(actions, dtype=np.float64)
        B, H, A = actions.shape
        self.launches += 1
        target, W, W_T, kappa, clip, smin, smax = self._cost
        Js, gs = [], []
        for b in range(B):
            J, g, *_ = adjoint.lcb_and_gradient(self.f, actions[b], np.asarray(mu0), np.asarray(S0), target, W, W_T, kappa,
                                                include_time, time0)
            Js.append(J)
            gs.append(g.reshape(H, A))
        out = {"J": torch.as_tensor(np.array(Js)), "grad": torch.as_tensor(np.stack(gs))}
        if trajectories:
            out.update({k: v for k, v in self.rollout(actions, mu0, S0, include_time, time0).items() if k != "J"})
        keys = [k for k in ("J", "grad", "mu", "Sig", "cost_mu", "cost_var") if k in out]
        out["packed"] = torch.cat([out[k].reshape(-1) for k in keys])
        out["layout"], off = [], 0
        for k in keys:
            out["layout"].append((k, tuple(out[k].shape), off, out[k].numel()))
            off += out[k].numel()
        return out

    def objective_grad_host(self, actions, mu0, S0, include_time=False, time0=0.0):
        """HipEngine.objective_grad_host: one sequence (H, A), numpy in / numpy out."""
        out = self.rollout_grad(np.asarray(actions, dtype=np.float64)[None], mu0, S0, include_time, time0, trajectories=True)
        return {k: out[k].numpy().copy() for k in ("J", "grad", "mu", "Sig", "cost_mu", "cost_var")}

    @staticmethod
    def host_views(out):
        return {k: out["packed"][off:off + n].view(sh) for k, sh, off, n in out["layout"]}


class DryRunEngine:
    """`bench.py --dry-run`: the engine methods bench.py touches, with an objective that costs nothing (a fixed function of the
    action sequence) -- a multi-rank run of the bench on the CPU then exercises everything AROUND the launches at the full
    config-5 slice sizes (8192 candidates over 8 ranks), which the CPU oracle could not evaluate in hours."""
    last_rollout_path = 1
    last_cluster = 1
    last_prepare_mode = 1
    build_id = "dry-run"

    def __init__(self, D):
        self.device = torch.device("cpu")
        self.D = D
        self.launches = 0

    def set_option(self, name, value):
        pass

    def set_cost(self, *a, **k):
        pass

    def prepare(self, X, Y, *a):
        self.N = X.shape[0]

    def rollout(self, actions, mu0, S0, include_time=False, time0=0.0, trajectories=True, stage_costs=True, out=None):
        actions = torch.as_tensor(actions, dtype=torch.float64)
        B, H, A = actions.shape
        self.launches += 1
        if out is None:
            out = {"mu": torch.zeros((B, H + 1, self.D), dtype=torch.float64), "Sig": torch.zeros((B, H + 1, self.D, self.D), dtype=torch.float64),
                   "cost_mu": torch.zeros((B, H + 1), dtype=torch.float64), "cost_var": torch.zeros((B, H + 1), dtype=torch.float64),
                   "J": torch.zeros(B, dtype=torch.float64)}
            out["mu"][:] = torch.as_tensor(np.asarray(mu0))
            out["Sig"][:] = torch.as_tensor(np.asarray(S0))
        w = torch.cos(torch.arange(1, H * A + 1, dtype=torch.float64))
        out["J"].copy_(((actions.reshape(B, -1) - 0.5) ** 2 @ w.abs()) + 0.01 * (actions.reshape(B, -1) @ w))
        return out

    def rollout_timed(self, actions, mu0, S0, reps, include_time=False, time0=0.0):
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            out = self.rollout(actions, mu0, S0)
        return (time.perf_counter() - t0) / reps * 1e3, out["J"]

    argmin_async = OracleEngine.argmin_async

    def close(self):
        pass
