"""TEST INFRASTRUCTURE ONLY -- fp64 numpy restatement of the reference GP-MPC hot path.

Every function cites the reference lines it follows (paths relative to
/root/reference/rl_gp_mpc/).  The restatement keeps the reference's *formulation*
(B / t / R / Q / maha as written there, LU solves and determinants) but

* carries a leading candidate axis ``B`` (the reference evaluates one action
  sequence per call, control_objects/controllers/gp_mpc_controller.py:125-148), and
* never materialises the reference's ``(D, D, N, N)`` temporaries
  (control_objects/models/gp_model.py:166,169-171): it loops over output pairs
  ``a <= b`` and mirrors ``S_ab`` (the reference computes both triangles; they agree
  to rounding).

Parity pinning: ``tests/test_oracle_vs_golden.py`` checks this module against
fixtures in ``tests/golden/`` that were produced by executing the reference's own
code (tools/gen_golden.py).  The one piece that is NOT pinned against the real
library is the K(X, X) evaluation, which the reference delegates to gpytorch
(models/gp_model.py:391,425; gpytorch is unpinned upstream and absent here):
``rbf_ard_gram`` restates its published closed form
``outputscale * exp(-1/2 * sum_e ((x_e - x'_e) / l_e)^2)`` -- "K-build parity unpinned".
"""
import numpy as np


# --------------------------------------------------------------------------- a1
def rbf_ard_gram(X, lengthscales, outputscales):
    """ScaleKernel(RBFKernel(ard_num_dims=E)) closed form (models/gp_model.py:391,425).

    X (N,E), lengthscales (D,E), outputscales (D,) -> K (D,N,N) without noise.
    """
    D, E = lengthscales.shape
    N = X.shape[0]
    K = np.empty((D, N, N))
    for a in range(D):                                            # one (N,N) temporary at a time
        sq = np.zeros((N, N))
        for e in range(E):
            xe = X[:, e] / lengthscales[a, e]
            d = xe[:, None] - xe[None, :]
            sq += d * d
        K[a] = outputscales[a] * np.exp(-0.5 * sq)
    return K


def factorize(X, Y, lengthscales, outputscales, noises, K=None):
    """calculate_factorizations (models/gp_model.py:400-431).

    L = chol(K + noise I); iK = cholesky_solve(I, L); beta = cholesky_solve(y, L).
    Returns iK (D,N,N), beta (D,N).  No jitter, no retry (as the reference).
    """
    from scipy.linalg import cho_solve
    D = Y.shape[1]
    N = X.shape[0]
    if K is None:
        K = rbf_ard_gram(X, lengthscales, outputscales)
    iK = np.empty((D, N, N))
    beta = np.empty((D, N))
    eye = np.eye(N)
    for a in range(D):
        L = np.linalg.cholesky(K[a] + noises[a] * eye)             # :427
        iK[a] = cho_solve((L, True), eye)                          # :428
        beta[a] = cho_solve((L, True), Y[:, a])                    # :429-430
    return iK, beta


class Factors:
    """What prepare_inference caches (models/gp_model.py:182-191)."""

    def __init__(self, X, Y, lengthscales, outputscales, noises, iK=None, beta=None):
        self.X = np.asarray(X, dtype=np.float64)
        self.Y = np.asarray(Y, dtype=np.float64)
        self.lengthscales = np.asarray(lengthscales, dtype=np.float64)
        self.variances = np.asarray(outputscales, dtype=np.float64)
        self.noises = np.asarray(noises, dtype=np.float64)
        if iK is None or beta is None:
            iK, beta = factorize(self.X, self.Y, self.lengthscales, self.variances, self.noises)
        self.iK = iK
        self.beta = beta


# --------------------------------------------------------------------------- a3
def moment_match_step(f, m, s):
    """predict_next_state_change (models/gp_model.py:112-180), batched over candidates.

    m (B,E) input mean, s (B,E,E) input covariance ->
    M (B,D), S (B,D,D), V (B,E,D)   (the reference returns M.t(), S, V.t(): (1,D),(D,D),(E,D)).
    """
    X, ls, var, beta, iK = f.X, f.lengthscales, f.variances, f.beta, f.iK
    N, E = X.shape
    D = ls.shape[0]
    Bc = m.shape[0]
    eyeE = np.eye(E)
    inp = X[None, :, :] - m[:, None, :]                            # :138  (B,N,E)

    M = np.empty((Bc, D))
    V = np.empty((Bc, E, D))
    for a in range(D):
        iL = 1.0 / ls[a]                                           # diag of self.iL :191
        iN = inp * iL                                              # :140  (B,N,E)
        Bm = iL[None, :, None] * s * iL[None, None, :] + eyeE      # :141
        t = np.linalg.solve(Bm, iN.transpose(0, 2, 1)).transpose(0, 2, 1)   # :145-146
        lb = np.exp(-0.5 * np.sum(iN * t, axis=-1)) * beta[a]      # :148  (B,N)
        tiL = t * iL                                               # :149
        c = var[a] / np.sqrt(np.linalg.det(Bm))                    # :150  (B,)
        M[:, a] = np.sum(lb, axis=-1) * c                          # :152
        V[:, :, a] = np.einsum('bne,bn->be', tiL, lb) * c[:, None]  # :153

    logv = np.log(var)
    S = np.empty((Bc, D, D))
    k = [logv[a] - 0.5 * np.sum((inp / ls[a]) ** 2, axis=-1) for a in range(D)]   # :168 (B,N)
    for a in range(D):
        for b in range(a, D):
            R = s * (1.0 / ls[a] ** 2 + 1.0 / ls[b] ** 2)[None, None, :] + eyeE   # :156-159
            Q = np.linalg.solve(R, s) / 2.0                                     # :163
            Xa = inp / ls[a] ** 2                                               # :161  (B,N,E)
            Xb = -inp / ls[b] ** 2                                              # :162
            XaQ = Xa @ Q
            XbQ = Xb @ Q
            Xs = np.sum(XaQ * Xa, axis=-1)                                      # :164
            X2s = np.sum(XbQ * Xb, axis=-1)                                     # :165
            maha = -2.0 * (XaQ @ Xb.transpose(0, 2, 1)) + Xs[:, :, None] + X2s[:, None, :]   # :166
            Lm = np.exp(k[a][:, :, None] + k[b][:, None, :] + maha)             # :169 (B,N,N)
            sab = np.einsum('i,bij,j->b', beta[a], Lm, beta[b])                 # :170-171
            if a == b:
                sab = sab - np.einsum('ij,bij->b', iK[a], Lm)                   # :173-175
            sab = sab / np.sqrt(np.linalg.det(R))                               # :176
            if a == b:
                sab = sab + var[a]                                              # :177
            S[:, a, b] = sab
            S[:, b, a] = sab
    S = S - M[:, :, None] * M[:, None, :]                                       # :178
    return M, S, V


# --------------------------------------------------------------------------- a4
def predict_trajectory(f, actions, mu0, S0, include_time=False, time0=0.0):
    """predict_trajectory (models/gp_model.py:60-110), batched over candidates.

    actions (B,H,A); mu0 (D,); S0 (D,D) -> mu (B,H+1,D), Sig (B,H+1,D,D); index 0 = input.
    """
    actions = np.asarray(actions, dtype=np.float64)
    Bc, H, A = actions.shape
    D = f.lengthscales.shape[0]
    E = f.X.shape[1]
    mu = np.empty((Bc, H + 1, D))
    Sig = np.empty((Bc, H + 1, D, D))
    mu[:, 0] = mu0                                                 # :91
    Sig[:, 0] = S0                                                 # :92
    for t in range(1, H + 1):                                      # :95
        s = np.zeros((Bc, E, E))
        s[:, :D, :D] = Sig[:, t - 1]                               # :96-97
        m = np.empty((Bc, E))
        m[:, :D] = mu[:, t - 1]                                    # :99
        m[:, D:D + A] = actions[:, t - 1]                          # :100
        if include_time:
            m[:, -1] = time0 + t - 1                               # :101-102
        M, S, V = moment_match_step(f, m, s)                       # :103
        mu[:, t] = mu[:, t - 1] + M                                # :105
        C = s[:, :D, :] @ V                                        # :107  (B,D,D)
        Sig[:, t] = S + Sig[:, t - 1] + C + C.transpose(0, 2, 1)   # :106-108
    return mu, Sig


# --------------------------------------------------------------------------- a6
def _norm_cdf(x, mu, sigma):
    """normal_cdf (control_objects/utils/pytorch_utils.py:16-17)."""
    from scipy.special import erf
    return 0.5 * (1.0 + erf((x - mu) / (sigma * np.sqrt(2.0))))


def stage_costs(mu, Sig, actions, target, W, W_T, state_min=None, state_max=None):
    """get_rewards_trajectory (states_reward_mappers/setpoint_distance_reward_mapper.py:144-149)
    = get_reward on t=0..H-1 (:12-68) + get_reward_terminal on t=H (:124-142).

    Returns (cost_mu (B,H+1), cost_var (B,H+1)); the reference returns rewards = -cost_mu.
    state_min/state_max given => use_constraints branch (:58-66), including the
    reference's quirk of passing the VARIANCE diagonal where a std is expected.
    """
    Bc, H1, D = mu.shape
    H = H1 - 1
    A = actions.shape[-1]
    err = np.concatenate([mu[:, :H], actions], axis=-1) - target            # :36  (B,H,D+A)
    Sa = np.zeros((Bc, H, D + A, D + A))
    Sa[:, :, :D, :D] = Sig[:, :H]                                            # :37-44
    cm = np.trace(Sa @ W, axis1=-1, axis2=-2) + np.einsum('bhi,ij,bhj->bh', err, W, err)   # :47-51
    TS = W @ Sa                                                              # :52
    cv = np.trace(2.0 * TS @ TS, axis1=-1, axis2=-2) \
        + 4.0 * np.einsum('bhi,bhij,jk,bhk->bh', err, TS, W, err)            # :53-56
    if state_min is not None:
        dg = np.diagonal(Sig[:, :H], axis1=-1, axis2=-2)
        pmin = _norm_cdf(state_min, mu[:, :H], dg)                           # :60,63
        pmax = 1.0 - _norm_cdf(state_max, mu[:, :H], dg)                     # :61,64
        cm = cm + pmax.sum(-1) + pmin.sum(-1)                                # :66
    eT = mu[:, H] - target[:D]                                               # :135
    cmT = np.trace(Sig[:, H] @ W_T, axis1=-1, axis2=-2) + np.einsum('bi,ij,bj->b', eT, W_T, eT)   # :136-137
    TST = W_T @ Sig[:, H]                                                    # :138
    cvT = np.trace(2.0 * TST @ TST, axis1=-1, axis2=-2) \
        + 4.0 * np.einsum('bi,bij,jk,bk->b', eT, TST, W_T, eT)               # :139-141
    return np.concatenate([cm, cmT[:, None]], axis=1), np.concatenate([cv, cvT[:, None]], axis=1)


# --------------------------------------------------------------------------- a5
def lcb_objective(cost_mu, cost_var, kappa, clip_to_zero=False):
    """Forward value of compute_mean_lcb_trajectory
    (control_objects/controllers/gp_mpc_controller.py:269-276): J = -mean_t(r_t + kappa*sqrt(var_t)).
    """
    ucb = -cost_mu + kappa * np.sqrt(cost_var)                               # :270
    if clip_to_zero:
        ucb = np.minimum(ucb, 0.0)                                           # :272-274
    return -ucb.mean(axis=-1)                                                # :275-276


def evaluate_candidates(f, w, actions=None, clip_to_zero=False, state_min=None, state_max=None):
    """The `optimize=False` candidate loop + argmin (gp_mpc_controller.py:125-148)
    over a batch: returns dict(mu, Sig, cost_mu, cost_var, J, best).  Tie-break = first
    (strict `<`, :146); a NaN in slot 0 is adopted (:146) and then never displaced,
    because `x < nan` is False -- reference behaviour, reproduced.
    """
    actions = w.actions if actions is None else actions
    mu, Sig = predict_trajectory(f, actions, w.mu0, w.S0, w.include_time, w.time0)
    cm, cv = stage_costs(mu, Sig, actions, w.target, w.W, w.W_T, state_min, state_max)
    J = lcb_objective(cm, cv, w.kappa, clip_to_zero)
    return dict(mu=mu, Sig=Sig, cost_mu=cm, cost_var=cv, J=J, best=first_wins_argmin(J))


def first_wins_argmin(J):
    """Selection rule of gp_mpc_controller.py:146-148 on a vector of objective values."""
    best, val = None, np.inf
    for i, v in enumerate(J):
        if v < val or (best is None and np.isnan(v)):
            best, val = i, v
    return best
