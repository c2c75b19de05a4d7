"""TEST INFRASTRUCTURE ONLY -- the CPU baseline of record ("port").

The real reference cannot travel to the GPU box (its Python never ships and it needs gpytorch),
so the CPU number reported next to the GPU number is this *unfused* torch fp64 restatement,
which keeps the reference's cost structure on purpose:

  * one candidate action sequence per call, Python loop over the horizon
    (rl_gp_mpc/control_objects/models/gp_model.py:95-108),
  * every pairwise quantity materialised for ALL (a, b) output pairs at once as a
    (D, D, N, N) tensor (gp_model.py:166,169-171), E x E LU solves / determinants
    (:146,150,163,176),
  * default torch intra-op threading (the reference never sets it on this path).

It is validated against the reference-generated goldens in tests/test_oracle_vs_golden.py and
is never imported by the product package.
"""
import torch


class UnfusedTorchModel:
    def __init__(self, X, iK, beta, lengthscales, variances):
        f64 = torch.float64
        self.X = torch.as_tensor(X, dtype=f64)
        self.iK = torch.as_tensor(iK, dtype=f64)
        self.beta = torch.as_tensor(beta, dtype=f64)
        self.ls = torch.as_tensor(lengthscales, dtype=f64)
        self.var = torch.as_tensor(variances, dtype=f64)
        self.D, self.E = self.ls.shape
        self.inv_ls = 1.0 / self.ls                      # (D,E)
        self.inv_ls2 = self.inv_ls ** 2

    @staticmethod
    def factorize(K, noises, Y):
        """gp_model.py:426-431 given K (D,N,N)."""
        N = K.shape[1]
        eye = torch.eye(N, dtype=K.dtype).expand_as(K)
        L = torch.linalg.cholesky(K + noises[:, None, None] * eye)
        iK = torch.cholesky_solve(eye.contiguous(), L)
        beta = torch.cholesky_solve(Y.t()[:, :, None], L)[:, :, 0]
        return iK, beta

    def step(self, m, s):
        """gp_model.py:112-180 for one input distribution (m (E,), s (E,E))."""
        D, E = self.D, self.E
        eye = torch.eye(E, dtype=m.dtype)
        nu = (self.X - m)                                               # (N,E)
        scaled = nu[None] * self.inv_ls[:, None, :]                     # (D,N,E)      :140
        Bmat = self.inv_ls[:, :, None] * s[None] * self.inv_ls[:, None, :] + eye   # (D,E,E)  :141
        t = torch.linalg.solve(Bmat, scaled.transpose(1, 2)).transpose(1, 2)       # :146
        lb = torch.exp(-0.5 * (scaled * t).sum(-1)) * self.beta         # (D,N)        :148
        c = self.var / torch.sqrt(torch.linalg.det(Bmat))               # :150
        M = lb.sum(-1) * c                                              # (D,)         :152
        V = torch.einsum('dne,dn->de', t * self.inv_ls[:, None, :], lb) * c[:, None]   # (D,E) :153

        dsum = self.inv_ls2[:, None, :] + self.inv_ls2[None, :, :]      # (D,D,E)
        R = s[None, None] * dsum[:, :, None, :] + eye                   # (D,D,E,E)    :156-159
        Q = torch.linalg.solve(R, s.expand(D, D, E, E)) / 2.0           # :163
        Xa = (nu[None] * self.inv_ls2[:, None, :])[:, None].expand(D, D, -1, E)    # (D,D,N,E) :161
        Xb = (-nu[None] * self.inv_ls2[:, None, :])[None].expand(D, D, -1, E)      # :162
        XaQ = Xa @ Q
        XbQ = Xb @ Q
        maha = -2.0 * (XaQ @ Xb.transpose(-1, -2)) + (XaQ * Xa).sum(-1)[..., :, None] \
            + (XbQ * Xb).sum(-1)[..., None, :]                          # (D,D,N,N)    :164-166
        k = torch.log(self.var)[:, None] - 0.5 * (scaled ** 2).sum(-1)  # (D,N)        :168
        Lm = torch.exp(k[:, None, :, None] + k[None, :, None, :] + maha)            # (D,D,N,N) :169
        S = torch.einsum('ai,abij,bj->ab', self.beta, Lm, self.beta)    # :170-171
        diagL = torch.stack([Lm[a, a] for a in range(D)])               # (D,N,N)      :173-174
        S = S - torch.diag((self.iK * diagL).sum((1, 2)))               # :175
        S = S / torch.sqrt(torch.linalg.det(R))                         # :176
        S = S + torch.diag(self.var)                                    # :177
        S = S - M[:, None] * M[None, :]                                 # :178
        return M, S, V.t()                                              # V.t(): (E,D)

    def predict_trajectory(self, actions, mu0, S0, include_time=False, time0=0.0):
        """gp_model.py:60-110 for one action sequence (H,A)."""
        H, A = actions.shape
        D, E = self.D, self.E
        mus = torch.empty((H + 1, D), dtype=torch.float64)
        Sigs = torch.empty((H + 1, D, D), dtype=torch.float64)
        mus[0] = mu0
        Sigs[0] = S0
        for t in range(1, H + 1):
            s = torch.zeros((E, E), dtype=torch.float64)
            s[:D, :D] = Sigs[t - 1]
            m = torch.empty(E, dtype=torch.float64)
            m[:D] = mus[t - 1]
            m[D:D + A] = actions[t - 1]
            if include_time:
                m[-1] = time0 + t - 1
            M, S, V = self.step(m, s)
            mus[t] = mus[t - 1] + M
            C = s[:D] @ V
            Sigs[t] = S + Sigs[t - 1] + C + C.t()
        return mus, Sigs

    @staticmethod
    def lcb(mus, Sigs, actions, target, W, W_T, kappa):
        """setpoint_distance_reward_mapper.py:144-149 + gp_mpc_controller.py:270-276 (forward)."""
        H, A = actions.shape
        D = mus.shape[1]
        err = torch.cat((mus[:-1], actions), 1) - target
        Sa = torch.zeros((H, D + A, D + A), dtype=torch.float64)
        Sa[:, :D, :D] = Sigs[:-1]
        cm = torch.diagonal(Sa @ W, dim1=-1, dim2=-2).sum(-1) + torch.einsum('hi,ij,hj->h', err, W, err)
        TS = W @ Sa
        cv = torch.diagonal(2 * TS @ TS, dim1=-1, dim2=-2).sum(-1) + 4 * torch.einsum('hi,hij,jk,hk->h', err, TS, W, err)
        eT = mus[-1] - target[:D]
        cmT = torch.trace(Sigs[-1] @ W_T) + eT @ W_T @ eT
        TST = W_T @ Sigs[-1]
        cvT = torch.trace(2 * TST @ TST) + 4 * eT @ TST @ W_T @ eT
        cm = torch.cat((cm, cmT[None]))
        cv = torch.cat((cv, cvT[None]))
        return -(-cm + kappa * torch.sqrt(cv)).mean()


def time_rollouts(w, n_rollouts, factors=None):
    """Forward rollouts/s of the unfused CPU path on workload `w` (first candidates)."""
    import time
    import numpy as np
    from .gpmpc_oracle import Factors
    f = factors or Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    m = UnfusedTorchModel(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    mu0, S0, target, W, W_T = tt(w.mu0), tt(w.S0), tt(w.target), tt(w.W), tt(w.W_T)
    acts = tt(w.actions)
    B = acts.shape[0]
    with torch.no_grad():
        mus, Sigs = m.predict_trajectory(acts[0], mu0, S0, w.include_time, w.time0)      # warm-up
        t0 = time.perf_counter()
        J = []
        for i in range(n_rollouts):
            mus, Sigs = m.predict_trajectory(acts[i % B], mu0, S0, w.include_time, w.time0)
            J.append(float(m.lcb(mus, Sigs, acts[i % B], target, W, W_T, w.kappa)))
        dt = time.perf_counter() - t0
    return n_rollouts / dt, dt, np.array(J)


def time_gradients(w, n_evals, factors=None):
    """Objective + gradient evaluations/s of the unfused CPU path: the reference's per-evaluation cost with
    optimize=True (forward + autograd backward through predict_trajectory, gp_mpc_controller.py:268-285)."""
    import time
    import numpy as np
    from .gpmpc_oracle import Factors
    f = factors or Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    m = UnfusedTorchModel(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    mu0, S0, target, W, W_T = tt(w.mu0), tt(w.S0), tt(w.target), tt(w.W), tt(w.W_T)
    acts = tt(w.actions)
    B = acts.shape[0]

    def one(i):
        a = acts[i % B].clone().requires_grad_(True)
        mus, Sigs = m.predict_trajectory(a, mu0, S0, w.include_time, w.time0)
        J = m.lcb(mus, Sigs, a, target, W, W_T, w.kappa)
        (g,) = torch.autograd.grad(J, a)
        return float(J.detach()), g

    one(0)                                                                                 # warm-up
    t0 = time.perf_counter()
    for i in range(n_evals):
        one(i)
    dt = time.perf_counter() - t0
    return n_evals / dt, dt
