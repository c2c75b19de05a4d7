"""TEST INFRASTRUCTURE ONLY -- the hot path in x87 extended precision (numpy longdouble, 64-bit mantissa).

Why it exists ("what's weak" 1): the predicted covariance `S_ab` (reference
rl_gp_mpc/control_objects/models/gp_model.py:170-178) is an O(1e-5) remainder of N^2 terms of size ~1e2, so two
correct fp64 evaluations that round differently disagree by ~1e-11 absolute -- 1e-5 relative at N = 500.  To say
which of two fp64 results is the better one we need a value that is much closer to the exact result than either.
This module evaluates the SAME mathematical expressions as `gpmpc_oracle.py` (same citations) with every operation
-- K build, Cholesky, triangular inverse, small solves, exp, sums -- in longdouble: unit round-off 5.4e-20 instead
of 1.1e-16, i.e. the same conditioning at 2^-11 of the rounding noise.  Inputs are the fp64 fixture values (exact
in longdouble).  `tools/gen_truth.py` writes the results next to the reference-made goldens; its `--self-check`
bounds the extended result's own noise by re-evaluating with the memory points permuted (a different summation
order) and, for one small case, against a 50-digit mpmath evaluation.

numpy.linalg has no longdouble kernels, hence the hand-written Cholesky / triangular inverse / Gauss elimination.
"""
import numpy as np

LD = np.longdouble


def _ld(a):
    return np.asarray(a, dtype=LD)


def cholesky_lower(A):
    """Left-looking column Cholesky, every operation in longdouble.  A (N,N) symmetric positive definite."""
    A = _ld(A)
    N = A.shape[0]
    L = np.zeros((N, N), dtype=LD)
    for j in range(N):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        if not d > 0:
            raise np.linalg.LinAlgError(f"pivot {j} not positive")
        L[j, j] = np.sqrt(d)
        if j + 1 < N:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def lower_inverse(L):
    """Y = L^-1 by forward substitution on the identity (row i from rows < i)."""
    N = L.shape[0]
    Y = np.zeros((N, N), dtype=LD)
    for i in range(N):
        r = -(L[i, :i] @ Y[:i, :i + 1]) if i else np.zeros(1, dtype=LD)
        r[i] += 1
        Y[i, :i + 1] = r / L[i, i]
    return Y


def gram(X, lengthscales, outputscales):
    """gp_model.py:391,425 closed form, longdouble."""
    X, ls, var = _ld(X), _ld(lengthscales), _ld(outputscales)
    D, E = ls.shape
    N = X.shape[0]
    K = np.empty((D, N, N), dtype=LD)
    for a in range(D):
        sq = np.zeros((N, N), dtype=LD)
        for e in range(E):
            xe = X[:, e] / ls[a, e]
            d = xe[:, None] - xe[None, :]
            sq += d * d
        K[a] = var[a] * np.exp(-sq / 2)
    return K


def factorize(X, Y, lengthscales, outputscales, noises):
    """gp_model.py:400-431 in longdouble: iK = L^-T L^-1, beta = iK y."""
    K = gram(X, lengthscales, outputscales)
    D, N = K.shape[0], K.shape[1]
    Yl = _ld(Y)
    nz = _ld(noises)
    iK = np.empty((D, N, N), dtype=LD)
    beta = np.empty((D, N), dtype=LD)
    for a in range(D):
        L = cholesky_lower(K[a] + nz[a] * np.eye(N, dtype=LD))
        Yi = lower_inverse(L)
        iK[a] = Yi.T @ Yi
        beta[a] = Yi.T @ (Yi @ Yl[:, a])
    return iK, beta


def solve_det(A, Bm):
    """Gauss elimination with partial pivoting on one small system: returns (A^-1 Bm, det A)."""
    A = A.copy()
    Bm = Bm.copy()
    n = A.shape[0]
    det = LD(1)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            Bm[[k, p]] = Bm[[p, k]]
            det = -det
        det = det * A[k, k]
        for r in range(k + 1, n):
            fct = A[r, k] / A[k, k]
            A[r, k:] -= fct * A[k, k:]
            Bm[r] -= fct * Bm[k]
    X = np.empty_like(Bm)
    for k in range(n - 1, -1, -1):
        X[k] = (Bm[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X, det


class Factors:
    def __init__(self, X, Y, lengthscales, outputscales, noises):
        self.X = _ld(X)
        self.lengthscales = _ld(lengthscales)
        self.variances = _ld(outputscales)
        self.iK, self.beta = factorize(X, Y, lengthscales, outputscales, noises)


def moment_match_step(f, m, s):
    """gp_model.py:112-180 for ONE input (m (E,), s (E,E)) in longdouble -> M (D,), S (D,D), V (E,D).
    Same expressions, line for line, as gpmpc_oracle.moment_match_step."""
    X, ls, var, beta, iK = f.X, f.lengthscales, f.variances, f.beta, f.iK
    N, E = X.shape
    D = ls.shape[0]
    eye = np.eye(E, dtype=LD)
    inp = X - m[None, :]                                            # :138
    M = np.empty(D, dtype=LD)
    V = np.empty((E, D), dtype=LD)
    for a in range(D):
        iL = 1 / ls[a]
        iN = inp * iL                                               # :140
        Bm = iL[:, None] * s * iL[None, :] + eye                    # :141
        t, detB = solve_det(Bm, iN.T.copy())                        # :145-146
        t = t.T
        lb = np.exp(-np.sum(iN * t, axis=-1) / 2) * beta[a]         # :148
        c = var[a] / np.sqrt(detB)                                  # :150
        M[a] = np.sum(lb) * c                                       # :152
        V[:, a] = ((t * iL) * lb[:, None]).sum(axis=0) * c          # :149,153
    logv = np.log(var)
    k = [logv[a] - np.sum((inp / ls[a]) ** 2, axis=-1) / 2 for a in range(D)]   # :168
    S = np.empty((D, D), dtype=LD)
    for a in range(D):
        for b in range(a, D):
            R = s * (1 / ls[a] ** 2 + 1 / ls[b] ** 2)[None, :] + eye             # :156-159
            Q, detR = solve_det(R, s.copy())
            Q = Q / 2                                                           # :163
            Xa = inp / ls[a] ** 2                                               # :161
            Xb = -inp / ls[b] ** 2                                              # :162
            XaQ = Xa @ Q
            XbQ = Xb @ Q
            maha = -2 * (XaQ @ Xb.T) + np.sum(XaQ * Xa, -1)[:, None] + np.sum(XbQ * Xb, -1)[None, :]   # :164-166
            Lm = np.exp(k[a][:, None] + k[b][None, :] + maha)                   # :169
            sab = beta[a] @ (Lm @ beta[b])                                      # :170-171
            if a == b:
                sab = sab - np.sum(iK[a] * Lm)                                  # :173-175
            sab = sab / np.sqrt(detR)                                           # :176
            if a == b:
                sab = sab + var[a]                                              # :177
            S[a, b] = S[b, a] = sab
    S = S - M[:, None] * M[None, :]                                             # :178
    return M, S, V


def predict_trajectory(f, actions, mu0, S0, include_time=False, time0=0.0):
    """gp_model.py:60-110 for ONE action sequence (H, A) in longdouble -> mu (H+1,D), Sig (H+1,D,D)."""
    actions = _ld(actions)
    H, A = actions.shape
    D = f.lengthscales.shape[0]
    E = f.X.shape[1]
    mu = np.empty((H + 1, D), dtype=LD)
    Sig = np.empty((H + 1, D, D), dtype=LD)
    mu[0] = _ld(mu0)
    Sig[0] = _ld(S0)
    for t in range(1, H + 1):
        s = np.zeros((E, E), dtype=LD)
        s[:D, :D] = Sig[t - 1]
        m = np.empty(E, dtype=LD)
        m[:D] = mu[t - 1]
        m[D:D + A] = actions[t - 1]
        if include_time:
            m[-1] = LD(time0) + (t - 1)
        M, S, V = moment_match_step(f, m, s)
        mu[t] = mu[t - 1] + M
        C = s[:D, :] @ V
        Sig[t] = S + Sig[t - 1] + C + C.T
    return mu, Sig


def mp_single_step(X, Y, lengthscales, outputscales, noises, m, s, digits=50):
    """One moment-matched step at `digits` decimal digits with mpmath (tiny N only: pure-Python loops).
    Used by tools/gen_truth.py --self-check to show what the longdouble evaluation itself is worth."""
    import mpmath as mp
    mp.mp.dps = digits
    N, E = X.shape
    D = Y.shape[1]
    f = lambda v: mp.mpf(float(v))                                   # noqa: E731  fp64 inputs are exact
    Xm = [[f(X[i, e]) for e in range(E)] for i in range(N)]
    ls = [[f(lengthscales[a, e]) for e in range(E)] for a in range(D)]
    var = [f(v) for v in outputscales]
    mm = [f(v) for v in m]
    sm = mp.matrix([[f(s[i, j]) for j in range(E)] for i in range(E)])
    iK, beta = [], []
    for a in range(D):
        K = mp.matrix(N, N)
        for i in range(N):
            for j in range(N):
                q = sum(((Xm[i][e] - Xm[j][e]) / ls[a][e]) ** 2 for e in range(E))
                K[i, j] = var[a] * mp.e ** (-q / 2) + (f(noises[a]) if i == j else 0)
        Ki = K ** -1
        iK.append(Ki)
        beta.append(Ki * mp.matrix([f(Y[i, a]) for i in range(N)]))
    inp = [[Xm[i][e] - mm[e] for e in range(E)] for i in range(N)]
    Mv = [None] * D
    for a in range(D):
        Bm = mp.matrix(E, E)
        for i in range(E):
            for j in range(E):
                Bm[i, j] = sm[i, j] / (ls[a][i] * ls[a][j]) + (1 if i == j else 0)
        Bi = Bm ** -1
        tot = mp.mpf(0)
        for i in range(N):
            iN = mp.matrix([inp[i][e] / ls[a][e] for e in range(E)])
            tot += mp.e ** (-(iN.T * Bi * iN)[0] / 2) * beta[a][i]
        Mv[a] = tot * var[a] / mp.sqrt(mp.det(Bm))
    S = [[None] * D for _ in range(D)]
    for a in range(D):
        for b in range(a, D):
            R = mp.matrix(E, E)
            for i in range(E):
                for j in range(E):
                    R[i, j] = sm[i, j] * (1 / ls[a][j] ** 2 + 1 / ls[b][j] ** 2) + (1 if i == j else 0)
            Q = (R ** -1) * sm / 2
            ka = [mp.log(var[a]) - sum((inp[i][e] / ls[a][e]) ** 2 for e in range(E)) / 2 for i in range(N)]
            kb = [mp.log(var[b]) - sum((inp[i][e] / ls[b][e]) ** 2 for e in range(E)) / 2 for i in range(N)]
            tot = mp.mpf(0)
            for i in range(N):
                xa = mp.matrix([inp[i][e] / ls[a][e] ** 2 for e in range(E)])
                xaQ = xa.T * Q
                for j in range(N):
                    z = mp.matrix([inp[i][e] / ls[a][e] ** 2 + inp[j][e] / ls[b][e] ** 2 for e in range(E)])
                    Lij = mp.e ** (ka[i] + kb[j] + (z.T * Q * z)[0])
                    wgt = beta[a][i] * beta[b][j] - (iK[a][i, j] if a == b else 0)
                    tot += wgt * Lij
                del xaQ
            sab = tot / mp.sqrt(mp.det(R)) + (var[a] if a == b else 0) - Mv[a] * Mv[b]
            S[a][b] = S[b][a] = sab
    return np.array([float(v) for v in Mv]), np.array([[float(S[a][b]) for b in range(D)] for a in range(D)])
