"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the analytic gradient dJ/du of the LCB objective.

The reference obtains the gradient by torch autograd through predict_trajectory
(rl_gp_mpc/control_objects/controllers/gp_mpc_controller.py:277, `mean_cost.backward()`, and :285).
The HIP path computes it analytically in two launches; this file states the same algebra in numpy, one
candidate at a time, so that the kernels can be checked quantity by quantity:

  forward  (gp_model.py:112-180 in the state-block form of DESIGN.md section 3), which also accumulates,
           for every output pair (a, b), the moments of  p_ij = u_i + w_j  under the weights
           E_ij = (beta_ai beta_bj - [a = b] iK_a,ij) L_ij:
               W = sum E_ij,  P1 = sum E_ij p_ij,  P2 = sum E_ij p_ij p_ij^T,  Pe = sum E_ij (nu_ie/l_ae^2 + nu_je/l_be^2)
           All nu_i = x_i - m move together when the input mean m moves, and Z = R^-1 Sigma enters the
           exponent as 1/2 p^T Z p, so these moments are everything the reverse sweep needs from the N^2 work.
  reverse  sweep over t = H-1 .. 0 with D x D algebra and one O(N D^2) pass over the points for the mean part.

Checked against torch autograd through oracle/unfused_torch.py (the reference's op sequence) and against
the reference-generated gradient goldens in tests/test_oracle_vs_golden.py.  Never imported by the product.
"""
import numpy as np


class StepRecord:
    pass


def _pairs(D):
    return [(a, b) for a in range(D) for b in range(a, D)]


def forward_step(f, m, Sig):
    """One moment-matching step (gp_model.py:112-180) for m (E,), Sig (D,D) + the stored moments."""
    X, beta, iK = f.X, f.beta, f.iK
    ils2 = 1.0 / f.lengthscales ** 2
    var = f.variances
    N, E = X.shape
    D = ils2.shape[0]
    nu = X - m
    nus = nu[:, :D]
    r = StepRecord()
    r.m, r.Sig = m.copy(), Sig.copy()
    r.Ai, r.c, r.s0, r.s1 = [], np.empty(D), np.empty(D), np.empty((D, D))
    M = np.empty(D)
    V = np.empty((D, D))
    for a in range(D):
        A = Sig + np.diag(1.0 / ils2[a, :D])
        Ai = np.linalg.inv(A)
        q = np.einsum('id,de,ie->i', nus, Ai, nus) + (nu[:, D:] ** 2 * ils2[a, D:]).sum(1)
        lb = np.exp(-0.5 * q) * beta[a]
        c = var[a] / np.sqrt(np.linalg.det(A) * np.prod(ils2[a, :D]))
        r.Ai.append(Ai)
        r.c[a], r.s0[a], r.s1[a] = c, lb.sum(), lb @ nus
        M[a] = c * r.s0[a]
        V[:, a] = c * (Ai @ r.s1[a])
    S = np.zeros((D, D))
    r.pairs = {}
    for (a, b) in _pairs(D):
        dab = ils2[a, :D] + ils2[b, :D]
        R = Sig * dab[None, :] + np.eye(D)
        Ri = np.linalg.inv(R)
        Z = Ri @ Sig
        rdet = 1.0 / np.sqrt(np.linalg.det(R))
        u = nus * ils2[a, :D]
        w = nus * ils2[b, :D]
        ka = np.log(var[a]) - 0.5 * (nu ** 2 * ils2[a]).sum(1) + 0.5 * np.einsum('id,de,ie->i', u, Z, u)
        kb = np.log(var[b]) - 0.5 * (nu ** 2 * ils2[b]).sum(1) + 0.5 * np.einsum('id,de,ie->i', w, Z, w)
        L = np.exp(ka[:, None] + kb[None, :] + u @ Z @ w.T)
        T = np.outer(beta[a], beta[b]) - (iK[a] if a == b else 0.0)
        Eh = T * L
        rs, cs = Eh.sum(1), Eh.sum(0)
        cross = u.T @ Eh @ w
        pr = StepRecord()
        pr.W = Eh.sum()
        pr.P1 = rs @ u + cs @ w
        pr.P2 = (u.T * rs) @ u + (w.T * cs) @ w + cross + cross.T
        pr.Pe = (rs @ nu[:, D:]) * ils2[a, D:] + (cs @ nu[:, D:]) * ils2[b, D:]
        pr.Ri, pr.Z, pr.rdet, pr.dab = Ri, Z, rdet, dab
        r.pairs[(a, b)] = pr
        S[a, b] = S[b, a] = rdet * pr.W + (var[a] if a == b else 0.0)
    S -= np.outer(M, M)
    r.M, r.V = M, V
    C = Sig @ V
    return M, S + C + C.T, r


def cost_terms(mu, Sig, u, target, W, state_min=None, state_max=None):
    """get_reward (setpoint_distance_reward_mapper.py:12-68) value and its partials wrt (mu, Sig, u).

    u = None: terminal cost (:124-142).  Returns cm, cv, (dcm/dmu, dcm/dSig, dcm/du), (dcv/...)."""
    from scipy.special import erf
    D = mu.shape[0]
    A = 0 if u is None else u.shape[0]
    err = (np.concatenate([mu, u]) if A else mu) - target[:D + A]
    Sa = np.zeros((D + A, D + A))
    Sa[:D, :D] = Sig
    G = W @ Sa @ W
    cm = np.trace(Sa @ W) + err @ W @ err
    cv = 2.0 * np.trace(W @ Sa @ W @ Sa) + 4.0 * err @ G @ err
    dcm_dSa = W.T
    dcm_derr = (W + W.T) @ err
    dcv_dSa = 4.0 * G.T + 4.0 * np.outer(W.T @ err, W @ err)
    dcv_derr = 4.0 * (G + G.T) @ err
    dcm_dmu, dcm_dSig = dcm_derr[:D].copy(), dcm_dSa[:D, :D].copy()
    if state_min is not None and A:
        sq = np.diagonal(Sig)                       # the reference passes the variance where a std is expected (:60-64)
        phi = lambda z: np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)      # noqa: E731
        zmin, zmax = (state_min - mu) / sq, (state_max - mu) / sq
        cm = cm + (0.5 * (1.0 + erf(zmin / np.sqrt(2.0)))).sum() + (1.0 - 0.5 * (1.0 + erf(zmax / np.sqrt(2.0)))).sum()
        dcm_dmu = dcm_dmu + (-phi(zmin) + phi(zmax)) / sq
        dcm_dSig = dcm_dSig + np.diag((-phi(zmin) * zmin + phi(zmax) * zmax) / sq)
    return cm, cv, (dcm_dmu, dcm_dSig, dcm_derr[D:]), (dcv_derr[:D], dcv_dSa[:D, :D], dcv_derr[D:])


def backward_step(f, r, mu_bar_n, Sig_bar_n):
    """Adjoint of forward_step + state update: (mu_bar', Sig_bar') at t+1 -> (mu_bar, Sig_bar, m_bar) at t."""
    X, beta = f.X, f.beta
    ils2 = 1.0 / f.lengthscales ** 2
    N, E = X.shape
    D = ils2.shape[0]
    Sig, M, V = r.Sig, r.M, r.V
    nu = X - r.m
    nus = nu[:, :D]
    Sb = 0.5 * (Sig_bar_n + Sig_bar_n.T)
    Sig_bar = Sb.copy()
    m_bar = np.zeros(E)
    m_bar[:D] = mu_bar_n
    Cb = 2.0 * Sb                                    # C + C^T
    Sig_bar += Cb @ V.T
    Vb = Sig @ Cb
    Mb = mu_bar_n - 2.0 * Sb @ M                     # mu' = mu + M,  S -= M M^T
    for a in range(D):
        Ai, c, s0, s1 = r.Ai[a], r.c[a], r.s0[a], r.s1[a]
        vb = Vb[:, a]
        y = Ai @ s1
        cb = Mb[a] * s0 + vb @ y
        s0b = Mb[a] * c
        s1b = c * (Ai @ vb)
        Aib = c * np.outer(vb, s1)
        q = np.einsum('id,de,ie->i', nus, Ai, nus) + (nu[:, D:] ** 2 * ils2[a, D:]).sum(1)
        lb = np.exp(-0.5 * q) * beta[a]
        om = -0.5 * lb * (s0b + nus @ s1b)           # q_bar_i
        G1 = om @ nus
        G2 = (nus.T * om) @ nus
        Ge = om @ nu[:, D:]
        Aib = Aib + G2
        Aib = 0.5 * (Aib + Aib.T)
        m_bar[:D] -= s0 * s1b + 2.0 * Ai @ G1        # nu = x - m
        m_bar[D:] -= 2.0 * ils2[a, D:] * Ge
        Ab = -Ai @ Aib @ Ai - 0.5 * cb * c * Ai
        Sig_bar += Ab
    for (a, b), pr in r.pairs.items():
        sb = Sb[a, b] if a == b else 2.0 * Sb[a, b]
        Wb = sb * pr.rdet
        Rb = -0.5 * sb * pr.W * pr.rdet * pr.Ri.T
        Zb = 0.5 * Wb * pr.P2
        m_bar[:D] += Wb * (pr.P1 - pr.dab * (pr.Z @ pr.P1))
        m_bar[D:] += Wb * pr.Pe
        Sig_bar += pr.Ri.T @ Zb
        Rb += -pr.Ri.T @ Zb @ pr.Z.T
        Sig_bar += Rb * pr.dab[None, :]
    Sig_bar = 0.5 * (Sig_bar + Sig_bar.T)
    return m_bar[:D], Sig_bar, m_bar


def lcb_and_gradient(f, actions, mu0, S0, target, W, W_T, kappa, include_time=False, time0=0.0,
                     state_min=None, state_max=None):
    """J and dJ/du (H, A) of compute_mean_lcb_trajectory (gp_mpc_controller.py:229-285) for one candidate.

    clip_lower_bound_cost_to_0 does not change the gradient (the reference clamps the value only)."""
    H, A = actions.shape
    D = mu0.shape[0]
    E = f.X.shape[1]
    mu, Sig = np.asarray(mu0, float).copy(), np.asarray(S0, float).copy()
    recs, mus, Sigs = [], [mu], [Sig]
    for t in range(H):
        m = np.zeros(E)
        m[:D] = mu
        m[D:D + A] = actions[t]
        if include_time:
            m[-1] = time0 + t
        M, dS, r = forward_step(f, m, Sig)
        recs.append(r)
        mu, Sig = mu + M, Sig + dS
        mus.append(mu)
        Sigs.append(Sig)
    n = H + 1
    J = 0.0
    grad = np.zeros((H, A))
    cmT, cvT, dmT, dvT = cost_terms(mus[H], Sigs[H], None, target, W_T)
    J += (cmT - kappa * np.sqrt(cvT)) / n
    wv = -kappa / (2.0 * np.sqrt(cvT))
    mu_bar = (dmT[0] + wv * dvT[0]) / n
    Sig_bar = (dmT[1] + wv * dvT[1]) / n
    for t in range(H - 1, -1, -1):
        mu_bar, Sig_bar, m_bar = backward_step(f, recs[t], mu_bar, Sig_bar)
        grad[t] = m_bar[D:D + A]
        cm, cv, dm, dv = cost_terms(mus[t], Sigs[t], actions[t], target, W, state_min, state_max)
        J += (cm - kappa * np.sqrt(cv)) / n
        wv = -kappa / (2.0 * np.sqrt(cv))
        mu_bar = mu_bar + (dm[0] + wv * dv[0]) / n
        Sig_bar = Sig_bar + 0.5 * ((dm[1] + wv * dv[1]) + (dm[1] + wv * dv[1]).T) / n
        grad[t] += (dm[2] + wv * dv[2]) / n
    return J, grad, np.array(mus), np.array(Sigs), recs
