"""Tier 1 (CPU): the oracle restatement vs goldens produced by the reference's own code.

Tolerances are fp64 reformulation noise amplified by cond(K + noise I) ~ 1e6:
factor quantities 1e-9 (scale-relative), propagated means 1e-8.  Covariances carry the
algorithm's own fp64 noise floor: S_ab is an O(1e-5) remainder of N^2 terms of size
|beta_i beta_j L_ij| ~ 1e2, so two correct fp64 evaluations that round the exponent
differently disagree by ~1e-11 absolute, i.e. up to ~1e-5 relative at N = 500
(measured: reference vs this restatement 1.1e-5 on traj_c3, 1.7e-7 on traj_c2).
SIG_TOL records that floor per case.
"""
import numpy as np
import pytest

from oracle import gpmpc_oracle as orc
from helpers import load, workload_of, factors_of, rel_err

SIG_TOL = {"traj_c3": 2e-5, "traj_c2": 2e-6, "traj_c4": 5e-6, "traj_c4_n1000": 5e-6}          # default 1e-7

TRAJ = ["traj_c1", "traj_c2", "traj_c3", "traj_c4", "traj_c4_n1000", "traj_c4_time", "traj_c5class", "traj_n1_dummy",
        "traj_clip", "traj_constraints", "traj_bigvar"]


@pytest.mark.parametrize("name", ["factor_n50", "factor_n96_d2"])
def test_factorisation(name):
    g = load(name)
    iK, beta = orc.factorize(g["X"], g["Y"], g["lengthscales"], g["outputscales"], g["noises"])
    assert rel_err(iK, g["iK"]) < 1e-9
    assert rel_err(beta, g["beta"]) < 1e-9


@pytest.mark.parametrize("name", ["step_zero_var", "step_dense_var", "step_dense_var_time"])
def test_single_step(name):
    g = load(name)
    w = workload_of(g)
    f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises, iK=g["iK"], beta=g["beta"])
    M, S, V = orc.moment_match_step(f, g["in_mean"][None], g["in_var"][None])
    assert rel_err(M[0], g["M"].ravel()) < 1e-10
    assert rel_err(S[0], g["S"]) < 1e-6      # S ~ 1e-5 left over from O(1e2) terms: 1e-13 abs floor
    assert rel_err(V[0], g["V"]) < 1e-10
    assert g["V"].shape == (w.X.shape[1], w.Y.shape[1])      # (E, D), not the docstring's (Ns, Ns+Na)


def test_zero_variance_is_gp_posterior():
    """Known answer: Sigma = 0 => M = k*^T beta, diag S = var - k*^T iK k*."""
    g = load("step_zero_var")
    w = workload_of(g)
    x = g["in_mean"]
    for a in range(w.Y.shape[1]):
        ks = w.outputscales[a] * np.exp(-0.5 * np.sum(((w.X - x) / w.lengthscales[a]) ** 2, axis=1))
        assert abs(ks @ g["beta"][a] - g["M"].ravel()[a]) < 1e-12
        assert abs(w.outputscales[a] - ks @ g["iK"][a] @ ks - g["S"][a, a]) < 1e-10


@pytest.mark.parametrize("name", TRAJ)
def test_trajectory_and_costs(name):
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    assert rel_err(f.beta, g["beta"]) < 1e-9
    smin = g["state_min"] if bool(g["use_constraints"]) else None
    smax = g["state_max"] if bool(g["use_constraints"]) else None
    out = orc.evaluate_candidates(f, w, clip_to_zero=bool(g["clip"]), state_min=smin, state_max=smax)
    H = w.actions.shape[1]
    assert out["mu"].shape == g["mu"].shape == (w.actions.shape[0], H + 1, w.Y.shape[1])
    assert np.array_equal(out["mu"][:, 0], np.broadcast_to(w.mu0, out["mu"][:, 0].shape))
    assert rel_err(out["mu"], g["mu"]) < 1e-8
    tol = SIG_TOL.get(name, 1e-7)
    assert rel_err(out["Sig"], g["Sig"]) < tol
    assert rel_err(-out["cost_mu"], g["rewards"]) < 1e-8
    assert rel_err(out["cost_var"], g["reward_vars"]) < tol
    assert rel_err(out["J"], g["J"]) < 1e-7


@pytest.mark.parametrize("name", ["traj_c1", "traj_c2", "traj_c3", "traj_c4", "traj_c4_n1000"])
def test_extended_precision_fixture_is_consistent(name):
    """`<name>_truth.npz` (tools/gen_truth.py: the trajectory with every operation in longdouble).  The distances it
    records are those of the committed golden; means agree to 1e-10; the reference's fp64 covariances sit at the
    method's cancellation floor, which grows with N (7.8e-8 at N = 200 ... 9.4e-6 at N = 500, D = 2)."""
    g, t = load(name), load(name + "_truth")
    assert abs(rel_err(g["Sig"], t["Sig"]) - float(t["ref_err_Sig"])) < 1e-9 + 1e-3 * float(t["ref_err_Sig"])
    assert rel_err(g["mu"], t["mu"]) < 1e-10
    assert float(t["ref_err_Sig"]) < 2e-5 and float(t["oracle_err_Sig"]) < 2e-5
    # triangle inequality: the tolerance the reference-vs-implementation tests may need
    assert float(t["ref_vs_oracle_Sig"]) <= float(t["ref_err_Sig"]) + float(t["oracle_err_Sig"]) + 1e-12


def test_extended_precision_step_agrees_with_the_fp64_oracle_on_a_small_case():
    """oracle/extended_precision.py restates the same expressions; at N = 30 the fp64 noise is ~1e-9 of S."""
    from oracle import extended_precision as xp
    from oracle import synth
    w = synth.make_workload(30, 3, 1, 1, 1, seed=4, s0=1e-3)
    E = w.X.shape[1]
    m = np.concatenate([w.mu0, w.actions[0, 0]])
    s = np.zeros((E, E))
    G = np.random.default_rng(1).standard_normal((3, 3)) * 0.03
    s[:3, :3] = G @ G.T + 1e-4 * np.eye(3)
    f = factors_of(w)
    M, S, V = orc.moment_match_step(f, m[None], s[None])
    fx = xp.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    assert rel_err(f.beta, fx.beta.astype(np.float64)) < 1e-9
    Mx, Sx, Vx = xp.moment_match_step(fx, xp._ld(m), xp._ld(s))
    assert rel_err(M[0], Mx.astype(np.float64)) < 1e-11
    assert rel_err(S[0], Sx.astype(np.float64)) < 1e-7
    assert rel_err(V[0], Vx.astype(np.float64)) < 1e-10


def test_argmin_trace():
    g = load("argmin_trace")
    w = workload_of(g)
    f = factors_of(w)
    out = orc.evaluate_candidates(f, w, actions=g["cand_actions"])
    assert rel_err(out["J"], g["cand_J"]) < 1e-8
    assert np.array_equal(g["cand_actions"][out["best"]], g["best_actions"])
    # candidates are what numpy's legacy global RNG yields (action_init_functions.py:4-5)
    np.random.seed(int(g["np_seed"]))
    H, A = g["cand_actions"].shape[1:]
    regen = []
    for _ in range(int(g["restarts"])):
        np.random.uniform(low=0, high=1, size=(H, A))       # the unused init draw (gp_mpc_controller.py:130)
        regen.append(np.random.uniform(low=0, high=1, size=(H, A)))   # the evaluated draw (:143)
    regen = np.stack(regen)
    assert np.array_equal(regen, g["cand_actions"])


def test_first_wins_and_nan_rule():
    assert orc.first_wins_argmin(np.array([3.0, 1.0, 1.0, 2.0])) == 1
    assert orc.first_wins_argmin(np.array([np.nan, 1.0])) == 0          # reference keeps a leading NaN
    assert orc.first_wins_argmin(np.array([2.0, np.nan, 1.0])) == 2


def test_symmetric_psd_on_well_conditioned_data():
    g = load("traj_c1")
    Sig = g["Sig"]
    assert np.max(np.abs(Sig - Sig.transpose(0, 1, 3, 2))) < 1e-10
    assert np.linalg.eigvalsh(0.5 * (Sig + Sig.transpose(0, 1, 3, 2))).min() > 0


@pytest.mark.parametrize("name", ["traj_c1", "traj_c4_time", "traj_c5class"])
def test_unfused_torch_baseline_matches_reference(name):
    """The CPU baseline of record (oracle/unfused_torch.py) reproduces the reference goldens."""
    import torch
    from oracle.unfused_torch import UnfusedTorchModel
    g = load(name)
    w = workload_of(g)
    K = torch.as_tensor(orc.rbf_ard_gram(w.X, w.lengthscales, w.outputscales))
    iK, beta = UnfusedTorchModel.factorize(K, torch.as_tensor(w.noises), torch.as_tensor(w.Y))
    assert rel_err(beta.numpy(), g["beta"]) < 1e-9
    m = UnfusedTorchModel(w.X, iK, beta, w.lengthscales, w.outputscales)
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    for b in range(min(2, w.actions.shape[0])):
        mus, Sigs = m.predict_trajectory(tt(w.actions[b]), tt(w.mu0), tt(w.S0), w.include_time, w.time0)
        assert rel_err(mus.numpy(), g["mu"][b]) < 1e-9
        assert rel_err(Sigs.numpy(), g["Sig"][b]) < 1e-7
        J = m.lcb(mus, Sigs, tt(w.actions[b]), tt(w.target), tt(w.W), tt(w.W_T), w.kappa)
        assert abs(float(J) - g["J"][b]) < 1e-8 * max(1.0, abs(g["J"][b]))


# ----------------------------------------------------------------------------- analytic gradient
@pytest.mark.parametrize("name", ["lcb_grad_norm", "lcb_grad_deriv"])
def test_numpy_adjoint_matches_reference_autograd(name):
    """oracle/adjoint.py (the algebra of the gradient kernels) vs the gradients the reference's
    `mean_cost.backward()` produced (gp_mpc_controller.py:277); fp64, 1e-7 of the gradient's scale."""
    from oracle import adjoint
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    acts = g["actions_model"]
    for b in range(acts.shape[0]):
        J, grad, mus, Sigs, _ = adjoint.lcb_and_gradient(f, acts[b], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa,
                                                         w.include_time, w.time0)
        assert abs(J - g["J"][b]) < 1e-9 * abs(g["J"][b])
        if bool(g["limit_action_change"]):
            # model[t] = clamp(prev + sum_{s<=t} (2 m u[s] - m)), clamp backward = identity (derivative_action_mapper.py:28-35)
            tail = np.cumsum(grad[::-1], axis=0)[::-1]
            grad = 2.0 * g["max_change"] * tail
        assert rel_err(grad.reshape(-1), g["grad"][b]) < 1e-7


@pytest.mark.parametrize("name", ["optimize_trace", "optimize_trace_deriv"])
def test_numpy_adjoint_replays_the_reference_optimize_true_trace(name):
    """Every point the reference's L-BFGS-B asked for in its `optimize=True` run (tools/gen_golden.py
    optimize_trace_case): the oracle's objective and gradient at that point vs what the reference computed there."""
    from oracle import adjoint
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    H, A = w.actions.shape[1:]
    for k in range(len(g["eval_J"])):
        u = g["eval_x"][k].reshape(H, A)
        if bool(g["limit_action_change"]):           # derivative_action_mapper.py:28-35 (clamp transparent here or not: value only)
            acts = np.clip(g["action_prev"] + np.cumsum(2.0 * g["max_change"] * u - g["max_change"], axis=0), 0.0, 1.0)
        else:
            acts = u
        J, grad, *_ = adjoint.lcb_and_gradient(f, acts, w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa, w.include_time, w.time0)
        assert abs(J - g["eval_J"][k]) < 1e-9 * abs(g["eval_J"][k]), k
        if bool(g["limit_action_change"]):
            grad = 2.0 * g["max_change"] * np.cumsum(grad[::-1], axis=0)[::-1]
        assert rel_err(grad.reshape(-1), g["eval_grad"][k]) < 1e-6, k


@pytest.mark.parametrize("N,D,A,H,tm", [(30, 3, 1, 5, False), (25, 2, 2, 4, True), (40, 4, 2, 3, False),
                                        (24, 9, 2, 2, False), (20, 16, 4, 2, True)])      # D > 8: beyond the gradient kernels today
def test_numpy_adjoint_matches_torch_autograd_of_the_reference_op_sequence(N, D, A, H, tm):
    import torch
    from oracle import adjoint, synth
    from oracle.unfused_torch import UnfusedTorchModel
    w = synth.make_workload(N, D, A, H, 2, include_time=tm, seed=1)
    f = factors_of(w)
    J, grad, mus, Sigs, _ = adjoint.lcb_and_gradient(f, w.actions[0], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa,
                                                     w.include_time, w.time0)
    m = UnfusedTorchModel(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)   # noqa: E731
    acts = tt(w.actions[0]).requires_grad_(True)
    mu_t, Sig_t = m.predict_trajectory(acts, tt(w.mu0), tt(w.S0), w.include_time, w.time0)
    Jt = m.lcb(mu_t, Sig_t, acts, tt(w.target), tt(w.W), tt(w.W_T), w.kappa)
    Jt.backward()
    assert abs(J - Jt.item()) < 1e-10 * abs(J)
    assert rel_err(grad, acts.grad.numpy()) < 1e-8
    assert rel_err(mus, mu_t.detach().numpy()) < 1e-10


def test_numpy_adjoint_with_state_constraints_matches_differences_of_the_oracle():
    """use_constraints adds the normal-cdf penalties (setpoint_distance_reward_mapper.py:58-66) to the stage cost;
    their partials are checked against central differences of the oracle's own forward objective."""
    from oracle import adjoint
    g = load("traj_constraints")
    w = workload_of(g)
    f = factors_of(w)
    smin, smax = g["state_min"], g["state_max"]
    a0 = w.actions[0]
    J, grad, *_ = adjoint.lcb_and_gradient(f, a0, w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa, w.include_time, w.time0,
                                           state_min=smin, state_max=smax)
    ref = orc.evaluate_candidates(f, w, actions=a0[None], state_min=smin, state_max=smax)
    assert abs(J - ref["J"][0]) < 1e-10 * abs(J)
    h = 1e-3                                        # 4th-order stencil: truncation ~ h^4, rounding ~ 1e-13 / h
    fd = np.zeros_like(a0)

    def Jof(a):
        return orc.evaluate_candidates(f, w, actions=a[None], state_min=smin, state_max=smax)["J"][0]
    for t in range(a0.shape[0]):
        for k in range(a0.shape[1]):
            d = np.zeros_like(a0)
            d[t, k] = h
            fd[t, k] = (8.0 * (Jof(a0 + d) - Jof(a0 - d)) - (Jof(a0 + 2 * d) - Jof(a0 - 2 * d))) / (12.0 * h)
    assert rel_err(grad, fd) < 1e-6


# ----------------------------------------------------------------------------- training loss
def test_training_loss_closed_form_gradient_matches_autograd():
    """oracle/gp_training.py: the exact-MLL loss of the training loop (gp_model.py:262-275; parity-unpinned against a
    gpytorch run, see the file header) -- closed-form gradient vs torch autograd of the same expression."""
    import torch
    from oracle import gp_training, synth
    w = synth.make_workload(40, 3, 1, 3, 2, seed=9)
    for a in range(3):
        loss, g_ls, g_os, g_nz = gp_training.neg_mll_and_grad(w.X, w.Y[:, a], w.lengthscales[a], w.outputscales[a], w.noises[a])
        tt = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True)   # noqa: E731
        ls, osc, nz = tt(w.lengthscales[a]), tt(w.outputscales[a]), tt(w.noises[a])
        lt = gp_training.neg_mll_torch(torch.as_tensor(w.X), torch.as_tensor(w.Y[:, a]), ls, osc, nz)
        lt.backward()
        assert abs(loss - lt.item()) < 1e-10 * abs(loss)
        assert rel_err(g_ls, ls.grad.numpy()) < 1e-8
        assert abs(g_os - osc.grad.item()) < 1e-8 * abs(g_os) and abs(g_nz - nz.grad.item()) < 1e-8 * abs(g_nz)


def test_training_loss_matches_independent_third_party_log_densities():
    """An independent pin for the loss the reference delegates to gpytorch (gp_model.py:262-275).  gpytorch's
    ExactMarginalLogLikelihood wraps `MultivariateNormal(mean, K + noise I).log_prob(y) / num_data`; gpytorch itself is
    absent from this image, but two third-party implementations of that log-density are not: torch.distributions (the
    class gpytorch's MultivariateNormal extends) and scipy.stats.  Both must give oracle/gp_training.py's loss.
    (Parity stays 'unpinned' by the rule -- no reference RUN produced these numbers -- but the restatement is no longer
    checked against itself only.)"""
    import torch
    from scipy.stats import multivariate_normal
    from oracle import gp_training, synth
    for N, seed in ((40, 9), (120, 3)):
        w = synth.make_workload(N, 3, 1, 3, 2, seed=seed)
        K = orc.rbf_ard_gram(w.X, w.lengthscales, w.outputscales)
        for a in range(3):
            loss = gp_training.neg_mll_and_grad(w.X, w.Y[:, a], w.lengthscales[a], w.outputscales[a], w.noises[a])[0]
            cov = K[a] + w.noises[a] * np.eye(N)
            mvn = torch.distributions.MultivariateNormal(torch.zeros(N, dtype=torch.float64), covariance_matrix=torch.as_tensor(cov))
            lp_torch = float(mvn.log_prob(torch.as_tensor(w.Y[:, a]))) / N
            lp_scipy = float(multivariate_normal.logpdf(w.Y[:, a], mean=np.zeros(N), cov=cov)) / N
            assert abs(loss + lp_torch) < 1e-10 * abs(loss), (loss, lp_torch)
            assert abs(loss + lp_scipy) < 1e-9 * abs(loss), (loss, lp_scipy)       # scipy factorises by eigendecomposition


def test_gram_matrix_matches_the_expanded_distance_form():
    """K(X, X) (gp_model.py:391,425 -- the one gpytorch call on the hot path) evaluated a second, differently ordered way:
    gpytorch's RBFKernel scales the inputs by 1/lengthscale, centres them, forms ||x||^2 + ||x'||^2 - 2 x.x' by a matrix
    product, zeroes the diagonal, clamps at 0 and applies exp(-d/2); ScaleKernel multiplies by the outputscale.  The
    oracle's direct-difference form must agree with that to fp64 rounding of the expanded form (~1e-14 absolute on d)."""
    from oracle import synth
    w = synth.make_workload(150, 4, 2, 3, 2, seed=5)
    K = orc.rbf_ard_gram(w.X, w.lengthscales, w.outputscales)
    for a in range(4):
        x = w.X / w.lengthscales[a]
        x = x - x.mean(axis=0)
        n2 = (x * x).sum(axis=1)
        d = n2[:, None] + n2[None, :] - 2.0 * (x @ x.T)
        np.fill_diagonal(d, 0.0)
        Ke = w.outputscales[a] * np.exp(-0.5 * np.maximum(d, 0.0))
        assert np.max(np.abs(Ke - K[a])) < 1e-13 * w.outputscales[a] * 50
        # and what that difference becomes after the factorisation (cond ~ 1e6): far inside the 1e-5 north-star bound
        iK1, b1 = orc.factorize(w.X, w.Y[:, a:a + 1], w.lengthscales[a:a + 1], w.outputscales[a:a + 1], w.noises[a:a + 1])
        iK2, b2 = orc.factorize(w.X, w.Y[:, a:a + 1], w.lengthscales[a:a + 1], w.outputscales[a:a + 1], w.noises[a:a + 1], K=Ke[None])
        assert rel_err(b2, b1) < 1e-7 and rel_err(iK2, iK1) < 1e-7
