"""Taylor degree K of the cross factor exp(g_i . w_j) per (output pair, horizon step): the bound |g . w| <= sum |Z_dd'| umax_d wmax_d'
taken with the data range measured from the INPUT MEAN (what the kernels do: csrc/rollout_kernel.h P1) against the range
measured from the centre of the data box (nu = (x - c) + (c - m), the cross terms fold into the per-point
factors, the range becomes half the box width).  CPU only (the numpy oracle supplies mu_t, Sigma_t).
  python tools/taylor_degree_table.py c2 [candidates] | c1 | c3 | c4 | oracle_c5_h50 | oracle_c5_inrange
K = 99 stands for "beyond the Taylor range" (tabulated / direct exponential).  Output kept in profiles/r05_taylor_degree_table.txt."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, gpmpc_oracle as orc
from math import factorial
KMAXARG = []
for K in range(15):
    # largest c with c^(K+1)/(K+1)! exp(2c) <= 2^-54
    lo, hi = 0.0, 5.0
    for _ in range(200):
        mid = 0.5*(lo+hi)
        if mid**(K+1)/factorial(K+1)*np.exp(2*mid) <= 2.0**-54: lo = mid
        else: hi = mid
    KMAXARG.append(lo)
print("thresholds", ["%.3g"%x for x in KMAXARG])
def degree(c):
    if c > KMAXARG[14]: return 99
    K = 1
    for k in range(1, 14): K += c > KMAXARG[k]
    return K
def cmax_pair(S, ils_a, ils_b, rg):
    D = S.shape[0]
    R = S * (ils_a[:D] + ils_b[:D])[None, :] + np.eye(D)
    Z = np.linalg.solve(R, S)
    return float(np.sum(np.abs(Z) * np.outer(rg * ils_a[:D], rg * ils_b[:D])))
def table(name, X, ls, mus, Sigs, D):
    ils = 1.0 / ls**2
    xmin, xmax = X.min(0)[:D], X.max(0)[:D]
    half = 0.5*(xmax - xmin)
    rows = []
    from collections import Counter
    hist_old, hist_new = Counter(), Counter()
    for t in range(mus.shape[0]):
        m, S = mus[t], Sigs[t]
        rg_old = np.maximum(np.abs(xmin - m), np.abs(xmax - m))
        for a in range(D):
            for b in range(a, D):
                co = cmax_pair(S, ils[a], ils[b], rg_old); cn = cmax_pair(S, ils[a], ils[b], half)
                hist_old[degree(co)] += 1; hist_new[degree(cn)] += 1
                rows.append((t, a, b, co, cn))
    r = np.array(rows)
    if "--by-step" in sys.argv:
        # K per (pair, step) for the first trajectory of the set: one line per horizon step, pairs in (a, b) order, "old>new"
        H1 = int(os.environ.get("KTABLE_STEPS", "0")) or len({int(x[0]) for x in rows})
        P = D * (D + 1) // 2
        print(f"# {name}: Taylor degree per (pair, step), bound from the input mean > bound from the box centre; 99 = beyond the Taylor range; first trajectory")
        for t in range(min(H1, len(rows) // P)):
            cells = []
            for q in range(P):
                _, a, b, co, cn = rows[t * P + q]
                ko, kn = degree(co), degree(cn)
                cells.append(f"{ko}>{kn}" if ko != kn else f"{ko}")
            print(f"t={t:2d} " + " ".join(cells))
    print(name, "pairs*steps", len(rows), "cmax old median %.3g max %.3g | new median %.3g max %.3g | ratio median %.2f" % (
        np.median(r[:,3]), r[:,3].max(), np.median(r[:,4]), r[:,4].max(), np.median(r[:,3]/r[:,4])))
    print("  K hist old", sorted(hist_old.items())); print("  K hist new", sorted(hist_new.items()))
    so = sum(k*v for k,v in hist_old.items() if k<99); sn = sum(k*v for k,v in hist_new.items() if k<99)
    print("  mean K (Taylor pairs) old %.2f new %.2f" % (so/max(1,sum(v for k,v in hist_old.items() if k<99)), sn/max(1,sum(v for k,v in hist_new.items() if k<99))))
which = sys.argv[1]
if which in ("c1","c2","c3","c4"):
    N,D,A,H,B,tm = synth.SHAPES[which]
    nb = int(sys.argv[2]) if len(sys.argv)>2 else 16
    w = synth.make_workload(N,D,A,H,max(B,nb),include_time=tm,seed=0)
    f = orc.Factors(w.X,w.Y,w.lengthscales,w.outputscales,w.noises)
    tr = orc.predict_trajectory(f, w.actions[:nb], w.mu0, w.S0)
    mu, Sig = tr[0], tr[1]
    print(mu.shape, Sig.shape)
    table(which, w.X, w.lengthscales, mu[:, :-1].reshape(-1, D), Sig[:, :-1].reshape(-1, D, D), D)
else:
    fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", which + ".npz"))
    N,D,A,H = int(fx["N"]),int(fx["D"]),int(fx["A"]),int(fx["H"])
    kw = dict(dynamics="contracting", dense_s0=0.05) if "inrange" in which else {}
    w = synth.make_workload(N,D,A,H,4,seed=int(fx["seed"]))
    table(which, w.X, w.lengthscales, fx["mu"][:, :-1].reshape(-1, D), fx["Sig"][:, :-1].reshape(-1, D, D), D)
