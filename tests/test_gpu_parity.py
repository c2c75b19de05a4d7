"""Tier 2 (GPU, through the C ABI): HIP kernels vs the goldens made from the reference's own
code and vs the CPU oracle on the same seeded inputs.

Tolerances (fp64): means 1e-8 scale-relative; covariances carry the algorithm's own fp64 noise
floor (see tests/test_oracle_vs_golden.py) -- two correct fp64 evaluations differ by up to ~1e-5
relative at N = 500 -- so SIG_TOL is per case and never looser than the north-star 1e-5 target (traj_c3 sits AT it:
the reference's own fp64 rounding is 9.4e-6 from the exact covariances there, so `_check_traj` also requires the HIP result to be
at least as close to the extended-precision truth as the reference is -- a summation-order change that moves |HIP - reference|
across 1e-5 then fails, or passes, for the right reason).
"""
import numpy as np
import pytest

from helpers import load, workload_of, factors_of, rel_err, record
from oracle import gpmpc_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

# covariance tolerances against the reference goldens, relative to max|Sigma|.  The north-star bound is 1e-5; config 3 sits just
# inside it because the REFERENCE's own fp64 rounding is 9.41e-6 away from the exact (extended-precision) covariances there (the HIP
# path: 4.3e-6, test_covariances_against_extended_precision) -- achieved |HIP - reference| 9.41e-6 on every build of rounds 3-5
# (profiles/*parity_report.json); the smaller memories are at their noise floor of 1e-7 ... 3e-7
SIG_TOL = {"traj_c3": 1e-5, "traj_c2": 1e-6, "traj_c4": 1e-6, "traj_c4_n1000": 1e-6}
TRAJ = ["traj_c1", "traj_c2", "traj_c3", "traj_c4", "traj_c4_n1000", "traj_c4_time", "traj_c5class", "traj_n1_dummy",
        "traj_clip", "traj_constraints", "traj_bigvar"]


@pytest.fixture(scope="module")
def engine():
    import gp_mpc_amd
    eng = gp_mpc_amd.HipEngine(0)
    yield eng
    eng.close()


def _set_cost(engine, w, g=None):
    clip = bool(g["clip"]) if g is not None and "clip" in g else False
    smin = g["state_min"] if g is not None and "use_constraints" in g and bool(g["use_constraints"]) else None
    smax = g["state_max"] if smin is not None else None
    engine.set_cost(w.target, w.W, w.W_T, w.kappa, clip, smin, smax)


def _check_traj(out, g, name, tag=None):
    tol = SIG_TOL.get(name, 1e-7)
    e_mu, e_S = rel_err(out["mu"].cpu().numpy(), g["mu"]), rel_err(out["Sig"].cpu().numpy(), g["Sig"])
    if tag:
        record(f"{tag}[{name}]", mu_vs_reference=e_mu, Sig_vs_reference=e_S, J_vs_reference=rel_err(out["J"].cpu().numpy(), g["J"]))
    assert e_mu < 1e-8
    assert e_S < tol
    if SIG_TOL.get(name, 0.0) >= 1e-5:
        # where the tolerance IS the north-star bound (no margin left): |HIP - exact| <= |reference - exact| (longdouble fixture,
        # tools/gen_truth.py) -- the reference's rounding, not this path's, is what fills the tolerance
        t = load(name + "_truth")
        e_truth = rel_err(out["Sig"].cpu().numpy(), t["Sig"])
        if tag:
            record(f"{tag}[{name}]", Sig_vs_truth=e_truth, reference_Sig_vs_truth=float(t["ref_err_Sig"]))
        assert e_truth <= float(t["ref_err_Sig"]), (e_truth, float(t["ref_err_Sig"]))
    assert rel_err(-out["cost_mu"].cpu().numpy(), g["rewards"]) < 1e-8
    assert rel_err(out["cost_var"].cpu().numpy(), g["reward_vars"]) < tol
    assert rel_err(out["J"].cpu().numpy(), g["J"]) < 1e-7


@pytest.mark.parametrize("name", TRAJ)
def test_rollout_with_reference_factors(engine, name):
    """Rollout kernel alone: factors come from the oracle (validated against the reference)."""
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    _set_cost(engine, w, g)
    out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    _check_traj(out, g, name, "rollout_with_reference_factors")


@pytest.mark.parametrize("name", TRAJ)
def test_prepare_then_rollout(engine, name):
    """Whole hot path on the GPU: K build + Cholesky + inverse + rollout + costs."""
    g = load(name)
    w = workload_of(g)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    iK, beta = engine.factors()
    assert rel_err(beta.cpu().numpy(), g["beta"]) < 1e-8
    _set_cost(engine, w, g)
    out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    _check_traj(out, g, name, "prepare_then_rollout")


@pytest.mark.parametrize("name", ["traj_c1", "traj_c2", "traj_c3", "traj_c4", "traj_c4_n1000"])
def test_covariances_against_extended_precision(engine, name):
    """Which fp64 result is right?  `<name>_truth.npz` (tools/gen_truth.py) holds the same trajectory evaluated with
    every operation in longdouble (own noise ~5e-9 on traj_c3, 1e-11 against 50-digit mpmath on a small case) and
    the distance of the reference's golden from it.  The reference's own fp64 evaluation is 9.4e-6 away from the
    exact covariances at N = 500 (config 3): that -- not an implementation error -- is why |HIP - reference| cannot
    be required below ~2e-5 there.  Required here: the HIP path (K build, factorisation and rollout on the GPU) is
    within the north-star 1e-5 of the EXACT result, and within 3x of what the two CPU fp64 evaluations (the
    reference's torch code, the numpy oracle) achieve -- they differ from each other by up to 2x at this level, the
    floor being rounding noise amplified by cond(K) ~ 1e6.  The achieved errors are written to the parity report."""
    g, t = load(name), load(name + "_truth")
    w = workload_of(g)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w, g)
    out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    e_mu, e_S = rel_err(out["mu"].cpu().numpy(), t["mu"]), rel_err(out["Sig"].cpu().numpy(), t["Sig"])
    record(f"extended_precision[{name}]", hip_mu=e_mu, hip_Sig=e_S, reference_mu=t["ref_err_mu"], reference_Sig=t["ref_err_Sig"],
           numpy_oracle_Sig=t["oracle_err_Sig"], hip_vs_reference_Sig=rel_err(out["Sig"].cpu().numpy(), g["Sig"]))
    print(f"{name}: |HIP - exact| mu {e_mu:.2e} Sig {e_S:.2e};  |reference - exact| mu {float(t['ref_err_mu']):.2e} "
          f"Sig {float(t['ref_err_Sig']):.2e}")
    assert e_mu < 1e-9
    assert e_S < 1e-5
    assert e_S <= 3.0 * max(float(t["ref_err_Sig"]), float(t["oracle_err_Sig"]), 1e-8)


@pytest.mark.parametrize("name", ["step_zero_var", "step_dense_var", "step_dense_var_time"])
@pytest.mark.parametrize("force_path", [0, 1, 2])
def test_single_step_known_answers(engine, name, force_path):
    """The reference's predict_next_state_change goldens (M, S, V for a zero and for a DENSE input covariance,
    gp_model.py:112-180) through the C ABI: an H = 1 rollout started from (mu0, Sigma0 = in_var[:D,:D]) returns
    mu_1 = mu_0 + M and Sigma_1 = S + Sigma_0 + C + C^T with C = Sigma_0 V[:D] (gp_model.py:105-108), which pins
    M, S and the state rows of V (the action / time rows of V are multiplied by the zero block of the input
    covariance in the reference too).  Dense Sigma_0 also covers what every trajectory golden lacks: a
    non-diagonal starting covariance."""
    g = load(name)
    w = workload_of(g)
    D = w.Y.shape[1]
    S0 = g["in_var"][:D, :D]
    engine.set_option("force_path", force_path)
    try:
        engine.set_factors(w.X, g["iK"], g["beta"], w.lengthscales, w.outputscales)
        _set_cost(engine, w)
        out = engine.rollout(w.actions[:1, :1], w.mu0, S0, w.include_time, w.time0)
    finally:
        engine.set_option("force_path", 0)
    M, S, V = g["M"].ravel(), g["S"], g["V"]
    C = S0 @ V[:D]
    mu1 = out["mu"].cpu().numpy()[0, 1]
    Sig1 = out["Sig"].cpu().numpy()[0, 1]
    e_M = float(np.max(np.abs(mu1 - w.mu0 - M)) / np.max(np.abs(M)))
    want = S + S0 + C + C.T
    e_S = float(np.max(np.abs(Sig1 - want)) / np.max(np.abs(want)))
    # the V term alone: Sigma_1 - S - Sigma_0 against C + C^T (only when Sigma_0 != 0)
    e_V = float(np.max(np.abs(Sig1 - S - S0 - C - C.T)) / max(np.max(np.abs(C)), 1e-300)) if np.any(S0) else 0.0
    record(f"single_step[{name},path{force_path}]", M=e_M, Sigma=e_S, V_term=e_V)
    assert e_M < 1e-9
    assert e_S < 1e-7
    assert e_V < 1e-5            # C ~ 1e-2 of Sigma_1 here: 1e-7 of Sigma_1 = 1e-5 of C


@pytest.mark.parametrize("name", ["factor_n50", "factor_n96_d2", "traj_c1"])
def test_factorisation_vs_reference(engine, name):
    g = load(name)
    engine.prepare(g["X"], g["Y"], g["lengthscales"], g["outputscales"], g["noises"])
    iK, beta = engine.factors()
    assert rel_err(iK.cpu().numpy(), g["iK"]) < 1e-8
    assert rel_err(beta.cpu().numpy(), g["beta"]) < 1e-8
    iKn = iK.cpu().numpy()
    assert np.array_equal(iKn, iKn.transpose(0, 2, 1))          # mirrored store: exactly symmetric


@pytest.mark.parametrize("N,D,A", [(1, 3, 1), (31, 2, 1), (33, 3, 1), (64, 1, 1), (240, 3, 1), (241, 3, 1), (256, 3, 1), (257, 4, 2), (300, 2, 1),
                                   (500, 2, 1), (512, 4, 2), (700, 2, 1)])
def test_factorisation_ragged_sizes(engine, N, D, A):
    """Panel width is 32: sizes below, at and across panel boundaries; N <= 240 takes the single-launch factorisation
    (one workgroup per GP; the measured crossover with the panel chain), larger memories the panel chain."""
    w = synth.make_workload(N, D, A, 3, 2, seed=N)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    iK, beta = engine.factors()
    iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    assert rel_err(iK.cpu().numpy(), iK0) < 1e-8
    assert rel_err(beta.cpu().numpy(), beta0) < 1e-8


@pytest.mark.parametrize("N,D,A", [(65, 2, 1), (257, 2, 1), (300, 3, 1), (352, 3, 2), (353, 2, 1), (400, 4, 2), (500, 2, 1), (544, 3, 1), (545, 2, 1), (639, 6, 2)])
def test_panel_chain_forms_agree_bit_for_bit(N, D, A):
    """The 32-wide panel path (240 < N < 640) in its forms (round 5): trailing update fused with the next diagonal block's factorisation
    (`prepare_fuse`), L^-1 in one launch after the factorisation (`prepare_invcols`: 1 = up to N = 352, 2 = up to 544), row blocks of
    L^-1 per side-stream launch (`prepare_inv_batch` 1 / 4), one or two streams (`prepare_overlap`) -- the same arithmetic in another
    launch structure: identical factors, and the oracle's to 1e-8.  Sizes at the form boundaries and with a ragged last panel."""
    import torch
    import gp_mpc_amd
    w = synth.make_workload(N, D, A, 3, 2, seed=N)
    iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng = gp_mpc_amd.HipEngine(0)
    try:
        eng.set_option("incremental", 0)
        eng.set_option("fused_prepare", 0)                    # the panel chain also below its crossover with the single-launch kernel
        ref = None
        for fuse, invcols, batch, overlap in [(0, 0, 1, 0), (0, 0, 1, 1), (1, 0, 1, 1), (1, 0, 4, 1), (1, 0, 3, 1), (1, 2, 1, 1), (1, 1, 4, 1), (0, 2, 4, 0)]:
            for k, v in (("prepare_fuse", fuse), ("prepare_invcols", invcols), ("prepare_inv_batch", batch), ("prepare_overlap", overlap)):
                eng.set_option(k, v)
            eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
            iK, beta = eng.factors()
            if ref is None:
                ref = (iK.clone(), beta.clone())
                assert rel_err(iK.cpu().numpy(), iK0) < 1e-8 and rel_err(beta.cpu().numpy(), beta0) < 1e-8
            else:
                assert torch.equal(iK, ref[0]) and torch.equal(beta, ref[1]), (fuse, invcols, batch, overlap)
    finally:
        eng.close()


LARGE = [(1024, 2, {}), (1025, 1, {}), (1151, 9, {}), (1300, 2, {}), (1300, 2, {"block128": 0}),
         (1300, 2, {"block128": 0, "inner_left": 0}), (1300, 2, {"tile128": 0}), (1300, 2, {"outer2": 0}),
         (1300, 2, {"outer2": 3}), (1300, 2, {"outer_block": 0}), (1700, 3, {"outer2": 1})]


@pytest.mark.parametrize("N,D,opts", LARGE, ids=[f"N{n}-D{d}-" + ("default" if not o else "-".join(f"{k}{v}" for k, v in o.items()))
                                                  for n, d, o in LARGE])
def test_large_memory_factorisation_paths(N, D, opts):
    """N >= 1024: outer panels of 128 columns (LDS-resident block factorisation, 128 x 128 tiled products, binary outer
    levels, triangular inverse by recursive doubling) -- sizes at and across the 128 / 256 / 512 boundaries, more GPs than
    XCDs, and every A/B option of the path, all against the CPU oracle's factorisation."""
    import gp_mpc_amd
    eng = gp_mpc_amd.HipEngine(0)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        w = synth.make_workload(N, D, 1, 2, 2, seed=N + D)
        eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        iK, beta = eng.factors()
        iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        e_iK, e_beta = rel_err(iK.cpu().numpy(), iK0), rel_err(beta.cpu().numpy(), beta0)
        record(f"large_factorisation[N{N},D{D},{opts}]", iK=e_iK, beta=e_beta)
        assert e_iK < 1e-8
        assert e_beta < 1e-8
        iKn = iK.cpu().numpy()
        assert np.array_equal(iKn, iKn.transpose(0, 2, 1))
        if not opts and D <= 2:
            # T_a (upper triangle, halved diagonal, zero rows after it) is only visible through a rollout
            _set_cost(eng, w)
            out = eng.rollout(w.actions[:1, :1], w.mu0, w.S0)
            f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises, iK0, beta0)
            mu, Sig = orc.predict_trajectory(f, w.actions[:1, :1], w.mu0, w.S0)
            assert rel_err(out["mu"].cpu().numpy(), mu) < 1e-8
            # S is an O(1e-6) remainder of N^2 terms of size ~1e2 here: two correct fp64 evaluations differ by ~3e-11
            # absolute; a wrong or misplaced T_a is an O(1) error
            assert rel_err(out["Sig"].cpu().numpy(), Sig) < 2e-4
    finally:
        eng.close()


def test_large_memory_not_positive_definite_then_recovers():
    """The block factorisation reports the first non-positive pivot (global index) and the handle stays usable."""
    import gp_mpc_amd
    eng = gp_mpc_amd.HipEngine(0)
    try:
        w = synth.make_workload(1100, 2, 1, 2, 2, seed=5)
        X = w.X.copy()
        X[900] = X[300]                      # duplicated point and negative noise => not positive-definite at pivot <= 901
        with pytest.raises(gp_mpc_amd.NotPositiveDefiniteError) as ei:
            eng.prepare(X, w.Y, w.lengthscales, w.outputscales, np.zeros(2) - 1e-3)
        assert "order" in str(ei.value)
        eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        iK, beta = eng.factors()
        iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        assert rel_err(iK.cpu().numpy(), iK0) < 1e-8
        assert rel_err(beta.cpu().numpy(), beta0) < 1e-8
    finally:
        eng.close()


def test_not_positive_definite_is_reported(engine):
    import gp_mpc_amd
    w = synth.make_workload(40, 2, 1, 3, 2, seed=3)
    w.X[7] = w.X[3]                      # duplicated point and zero noise => singular K
    with pytest.raises(gp_mpc_amd.NotPositiveDefiniteError):
        engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, np.zeros(2) - 1e-3)


@pytest.mark.parametrize("N0,D,A,tm", [(40, 3, 1, False), (200, 3, 1, False), (130, 4, 2, True), (1, 2, 1, False)])
def test_incremental_prepare_matches_full_factorisation(N0, D, A, tm):
    """gp_mpc_controller.py:117 refactorises every control step although the memory only grew by a
    point; gpmpc_prepare border-updates the cached inverse instead.  Same factors as a fresh
    factorisation of the grown memory (fp64, 1e-8 -- the Cholesky-vs-reference tolerance)."""
    import gp_mpc_amd
    steps = [1, 1, 3, 1, 8, 2]
    w = synth.make_workload(N0 + sum(steps), D, A, 4, 4, include_time=tm, seed=N0)
    inc, full = gp_mpc_amd.HipEngine(0), gp_mpc_amd.HipEngine(0)
    full.set_option("incremental", 0)
    try:
        n = N0
        inc.prepare(w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)
        assert inc.last_prepare_mode == 0
        for k in steps:
            n += k
            inc.prepare(w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)
            assert inc.last_prepare_mode == 1
        full.prepare(w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)
        assert full.last_prepare_mode == 0
        iK, beta = inc.factors()
        iK0, beta0 = full.factors()
        assert rel_err(iK.cpu().numpy(), iK0.cpu().numpy()) < 1e-8
        assert rel_err(beta.cpu().numpy(), beta0.cpu().numpy()) < 1e-8
        iKn = iK.cpu().numpy()
        assert np.array_equal(iKn, iKn.transpose(0, 2, 1))
        iKo, betao = orc.factorize(w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)
        assert rel_err(iKn, iKo) < 1e-8
        # and the rollouts that consume them agree
        for e in (inc, full):
            e.set_cost(w.target, w.W, w.W_T, w.kappa)
        a = inc.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        b = full.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        assert rel_err(a["mu"].cpu().numpy(), b["mu"].cpu().numpy()) < 1e-9
        assert rel_err(a["Sig"].cpu().numpy(), b["Sig"].cpu().numpy()) < 1e-6
        assert rel_err(a["J"].cpu().numpy(), b["J"].cpu().numpy()) < 1e-8
    finally:
        inc.close()
        full.close()


def test_prepare_reuse_rules():
    """Cache hit only for identical inputs; any change of an old point, of Y, of a hyper-parameter, a
    shrinking memory, more than 8 new points or the refresh interval force a full factorisation."""
    import gp_mpc_amd
    w = synth.make_workload(90, 3, 1, 3, 2, seed=5)
    e = gp_mpc_amd.HipEngine(0)
    try:
        args = lambda n: (w.X[:n], w.Y[:n], w.lengthscales, w.outputscales, w.noises)   # noqa: E731
        e.prepare(*args(60)); assert e.last_prepare_mode == 0
        e.prepare(*args(60)); assert e.last_prepare_mode == 2
        e.prepare(*args(61)); assert e.last_prepare_mode == 1
        e.prepare(*args(60)); assert e.last_prepare_mode == 0          # shrank
        e.prepare(*args(70)); assert e.last_prepare_mode == 0          # > 8 new points
        X2 = w.X[:71].copy(); X2[5, 0] += 1e-9
        e.prepare(X2, w.Y[:71], w.lengthscales, w.outputscales, w.noises); assert e.last_prepare_mode == 0
        Y2 = w.Y[:72].copy(); Y2[0, 1] += 1e-12
        e.prepare(X2[:71], Y2[:71], w.lengthscales, w.outputscales, w.noises); assert e.last_prepare_mode == 0
        e.prepare(*args(72)); assert e.last_prepare_mode == 0          # old point differs from the cache
        e.prepare(w.X[:73], w.Y[:73], w.lengthscales * 1.0000001, w.outputscales, w.noises); assert e.last_prepare_mode == 0
        e.prepare(w.X[:74], w.Y[:74], w.lengthscales * 1.0000001, w.outputscales, w.noises); assert e.last_prepare_mode == 1
        e.prepare(w.X[:75], w.Y[:75], w.lengthscales * 1.0000001, w.outputscales, w.noises * 2); assert e.last_prepare_mode == 0
        e.set_option("refresh_every", 2)
        modes = []
        for n in range(76, 82):
            e.prepare(w.X[:n], w.Y[:n], w.lengthscales * 1.0000001, w.outputscales, w.noises * 2)
            modes.append(e.last_prepare_mode)
        assert modes == [1, 1, 0, 1, 1, 0]
        iK, beta = e.factors()
        iKo, betao = orc.factorize(w.X[:81], w.Y[:81], w.lengthscales * 1.0000001, w.outputscales, w.noises * 2)
        assert rel_err(iK.cpu().numpy(), iKo) < 1e-8 and rel_err(beta.cpu().numpy(), betao) < 1e-8
        # set_factors drops the record: the next prepare factorises
        e.set_factors(w.X[:81], iKo, betao, w.lengthscales, w.outputscales)
        e.prepare(w.X[:81], w.Y[:81], w.lengthscales * 1.0000001, w.outputscales, w.noises * 2)
        assert e.last_prepare_mode == 0
    finally:
        e.close()


def test_incremental_update_reports_lost_positive_definiteness():
    import gp_mpc_amd
    w = synth.make_workload(41, 2, 1, 3, 2, seed=3)
    w.X[:40] += 50.0 * np.arange(40)[:, None]       # far apart: K ~ outputscale * I, PD even with noise < 0
    w.X[40] = w.X[3]                                 # the appended point duplicates an old one
    e = gp_mpc_amd.HipEngine(0)
    try:
        nz = -0.1 * np.asarray(w.outputscales)
        e.prepare(w.X[:40], w.Y[:40], w.lengthscales, w.outputscales, nz)
        with pytest.raises(gp_mpc_amd.NotPositiveDefiniteError):
            e.prepare(w.X, w.Y, w.lengthscales, w.outputscales, nz)
    finally:
        e.close()


@pytest.mark.parametrize("threads", [256, 512, 1024])
@pytest.mark.parametrize("force_global", [0, 1])
def test_rollout_variants_agree(engine, threads, force_global):
    """Workgroup sizes and the large-N (global scratch) variant give the same numbers."""
    g = load("traj_c1")
    w = workload_of(g)
    f = factors_of(w)
    engine.set_option("threads", threads)
    engine.set_option("force_global_scratch", force_global)
    try:
        engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
        _set_cost(engine, w, g)
        out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        _check_traj(out, g, "traj_c1")
    finally:
        engine.set_option("threads", 0)
        engine.set_option("force_global_scratch", 0)


@pytest.mark.parametrize("cols", [1, 2])
@pytest.mark.parametrize("force_path", [1, 2])
@pytest.mark.parametrize("name", ["traj_c1", "traj_c2", "traj_c4_time", "traj_n1_dummy", "traj_c5class"])
def test_one_and_two_columns_per_lane_agree_with_reference(engine, name, force_path, cols):
    """The pairwise pass with one column per lane and with two adjacent columns per lane (default for D <= 4), in the
    direct-exp and the element-wise Taylor form, against the reference goldens (odd N, N = 1, D = 6 included)."""
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    engine.set_option("cols_per_lane", cols)
    engine.set_option("force_path", force_path)
    try:
        engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
        _set_cost(engine, w, g)
        out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        _check_traj(out, g, name)
    finally:
        engine.set_option("cols_per_lane", 0)
        engine.set_option("force_path", 0)


@pytest.mark.parametrize("force_path", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["traj_c1", "traj_c2", "traj_c4", "traj_c4_time", "traj_c5class", "traj_bigvar", "traj_n1_dummy"])
def test_exp_taylor_and_separable_paths_agree_with_reference(engine, name, force_path):
    """0 = automatic (Taylor degree from the |g.w| bound; separable moments for off-diagonal pairs when
    D <= 4), 1 = always the direct exp(ka' + kb' + g.w) evaluation, 2 = Taylor but element-wise."""
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    engine.set_option("force_path", force_path % 3)
    engine.set_option("force_separable", int(force_path == 3))      # 3 = automatic degree, separable forced on
    try:
        engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
        _set_cost(engine, w, g)
        out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        _check_traj(out, g, name)
    finally:
        engine.set_option("force_path", 0)
        engine.set_option("force_separable", 0)


def test_large_input_variance_uses_high_degree_or_exp(engine):
    """Input covariances from 1e-8 to 0.3 sweep the Taylor degree up to the exp fallback; all three
    evaluation paths must agree with the CPU oracle."""
    for s0 in [1e-8, 1e-4, 1e-2, 1e-1, 3e-1]:
        w = synth.make_workload(90, 3, 1, 4, 6, seed=11, s0=s0, noise_var=1e-4)
        f = factors_of(w)
        ref = orc.evaluate_candidates(f, w)
        for force_path in (0, 1, 2, 3):
            engine.set_option("force_path", force_path % 3)
            engine.set_option("force_separable", int(force_path == 3))
            try:
                engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
                _set_cost(engine, w)
                out = engine.rollout(w.actions, w.mu0, w.S0)
            finally:
                engine.set_option("force_path", 0)
                engine.set_option("force_separable", 0)
            assert rel_err(out["mu"].cpu().numpy(), ref["mu"]) < 1e-9, (s0, force_path)
            assert rel_err(out["Sig"].cpu().numpy(), ref["Sig"]) < 1e-7, (s0, force_path)
            assert rel_err(out["J"].cpu().numpy(), ref["J"]) < 1e-8, (s0, force_path)


@pytest.mark.parametrize("name", ["traj_c5class", "traj_c4_time", "traj_c2"])
def test_large_n_variant_on_reference_goldens(engine, name):
    """The global-scratch (large-N) variant of the kernel, forced at small N, against the reference."""
    g = load(name)
    w = workload_of(g)
    f = factors_of(w)
    engine.set_option("force_global_scratch", 1)
    try:
        engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
        _set_cost(engine, w, g)
        out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        _check_traj(out, g, name)
    finally:
        engine.set_option("force_global_scratch", 0)


def test_config5_shape_class_runs_and_matches_oracle(engine):
    """D=16, E=20 at N=1024 (needs the global-scratch variant: per-point arrays exceed LDS), 2 steps;
    oracle on one candidate."""
    w = synth.make_workload(1024, 16, 4, 2, 8, seed=31)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    f = factors_of(w)
    assert rel_err(engine.factors()[1].cpu().numpy(), f.beta) < 1e-7
    ref = orc.evaluate_candidates(f, w, actions=w.actions[:1])
    assert rel_err(out["mu"].cpu().numpy()[:1], ref["mu"]) < 1e-8
    assert rel_err(out["Sig"].cpu().numpy()[:1], ref["Sig"]) < 1e-5
    assert rel_err(out["J"].cpu().numpy()[:1], ref["J"]) < 1e-7


@pytest.mark.parametrize("N,D,A,H,B,tm", [(70, 1, 1, 4, 3, False), (60, 5, 2, 3, 3, False), (48, 7, 1, 3, 2, True),
                                          (40, 3, 1, 1, 1, False), (33, 12, 3, 2, 2, False)])
def test_padded_and_odd_shapes_against_oracle(engine, N, D, A, H, B, tm):
    """State dimensions that are not compiled exactly run on the next padded size (D=1->2, 5->6, 7->8,
    12->16, runtime D < DP); plus H = 1, B = 1 and a time input."""
    w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=100 + D, time0=float(N) if tm else 0.0)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    f = factors_of(w)
    ref = orc.evaluate_candidates(f, w)
    for force_path in (0, 1):
        engine.set_option("force_path", force_path)
        try:
            out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        finally:
            engine.set_option("force_path", 0)
        assert rel_err(out["mu"].cpu().numpy(), ref["mu"]) < 1e-9, force_path
        # covariances: 1e-7 relative on top of the 1e-11 absolute cancellation floor of the method
        dS = np.max(np.abs(out["Sig"].cpu().numpy() - ref["Sig"]))
        assert dS < 1e-7 * np.max(np.abs(ref["Sig"])) + 2e-11, (force_path, dS)
        assert rel_err(out["cost_mu"].cpu().numpy(), ref["cost_mu"]) < 1e-9, force_path
        assert rel_err(out["cost_var"].cpu().numpy(), ref["cost_var"]) < 1e-5, force_path
        assert rel_err(out["J"].cpu().numpy(), ref["J"]) < 1e-8, force_path


@pytest.mark.parametrize("N,D,A,H,B,tm,s0", [(33, 8, 2, 3, 3, False, 1e-6), (100, 7, 1, 2, 2, True, 1e-3), (130, 12, 3, 2, 2, False, 1e-4),
                                            (17, 16, 4, 3, 2, False, 5e-2), (257, 16, 2, 2, 3, False, 1e-6), (64, 9, 1, 2, 1, False, 1e-2)])
def test_matrix_core_pair_pass_against_oracle(engine, N, D, A, H, B, tm, s0):
    """The streaming kernel's pairwise pass on the fp64 matrix cores (padded state dimension 8 or 16; config 5's path),
    forced at small N: ragged N (not a multiple of 16 / 64, N < 16), D below the padded size, time input, input
    variances that select low and high Taylor degrees and the direct-exp form, each in all evaluation modes (0 automatic,
    1 direct exp, 2 Taylor, 4 the tabulated mid-range form exp(c) = T[round(64 c)] P6(r) forced from |g.w| = 0 up)."""
    w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=300 + N, s0=s0, noise_var=1e-4, time0=float(N) if tm else 0.0)
    f = factors_of(w)
    ref = orc.evaluate_candidates(f, w)
    engine.set_option("force_global_scratch", 1)
    try:
        engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        _set_cost(engine, w)
        for force_path in (0, 1, 2, 4):
            engine.set_option("force_path", force_path)
            out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
            tag = (N, D, force_path)
            e_S = np.max(np.abs(out["Sig"].cpu().numpy() - ref["Sig"]))
            record(f"matrix_core_pair_pass[N{N},D{D},path{force_path}]", mu=rel_err(out["mu"].cpu().numpy(), ref["mu"]),
                   Sig_abs=e_S, J=rel_err(out["J"].cpu().numpy(), ref["J"]))
            assert rel_err(out["mu"].cpu().numpy(), ref["mu"]) < 1e-9, tag
            assert e_S < 1e-7 * np.max(np.abs(ref["Sig"])) + 2e-11, (tag, e_S)
            assert rel_err(out["J"].cpu().numpy(), ref["J"]) < 1e-8, tag
        again = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        assert np.array_equal(again["Sig"].cpu().numpy(), out["Sig"].cpu().numpy())          # fixed summation order
        for rows in (128, 256):              # the long row chunks used from N = 1024 up (8 / 16 row tiles per item)
            engine.set_option("rows_per_chunk", rows)
            engine.set_option("force_path", 0)
            big = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
            assert rel_err(big["mu"].cpu().numpy(), ref["mu"]) < 1e-9, rows
            assert np.max(np.abs(big["Sig"].cpu().numpy() - ref["Sig"])) < 1e-7 * np.max(np.abs(ref["Sig"])) + 2e-11, rows
    finally:
        engine.set_option("rows_per_chunk", 0)
        engine.set_option("force_path", 0)
        engine.set_option("force_global_scratch", 0)


def test_config5_full_size_step_against_oracle_fixture(engine):
    """BASELINE configs[4] at full size (N=4096, D=16, A=4, E=20): K build + factorisation + one
    moment-matched step for 2 candidates vs tests/golden/oracle_c5_step.npz (CPU oracle, tools/gen_golden_c5.py;
    the reference formulation cannot run this size)."""
    g = load("oracle_c5_step")
    w = synth.make_workload(int(g["N"]), int(g["D"]), int(g["A"]), int(g["H"]), int(g["B"]), seed=int(g["seed"]))
    assert np.allclose([w.X.sum(), w.Y.sum(), w.actions.sum()], g["x_checksum"], rtol=0, atol=1e-9)   # same inputs
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    assert rel_err(engine.factors()[1].cpu().numpy()[:, :64], g["beta_head"]) < 1e-6
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    assert rel_err(out["mu"].cpu().numpy(), g["mu"]) < 1e-8
    assert rel_err(out["Sig"].cpu().numpy(), g["Sig"]) < 1e-5
    assert rel_err(out["J"].cpu().numpy(), g["J"]) < 1e-7


def test_config5_full_size_five_steps_against_oracle_fixture(engine):
    """BASELINE configs[4] at full size, FIVE horizon steps (error growth over the recurrence at D = 16, N = 4096)
    for 2 candidates vs tests/golden/oracle_c5_traj.npz (CPU oracle, `tools/gen_golden_c5.py --steps 5`)."""
    g = load("oracle_c5_traj")
    w = synth.make_workload(int(g["N"]), int(g["D"]), int(g["A"]), int(g["H"]), int(g["B"]), seed=int(g["seed"]))
    assert np.allclose([w.X.sum(), w.Y.sum(), w.actions.sum()], g["x_checksum"], rtol=0, atol=1e-9)   # same inputs
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    mu, Sig = out["mu"].cpu().numpy(), out["Sig"].cpu().numpy()
    per_step = [rel_err(Sig[:, t], g["Sig"][:, t]) for t in range(1, int(g["H"]) + 1)]
    record("config5_five_steps", mu=rel_err(mu, g["mu"]), Sig=rel_err(Sig, g["Sig"]), J=rel_err(out["J"].cpu().numpy(), g["J"]),
           **{f"Sig_step{t + 1}": e for t, e in enumerate(per_step)})
    assert rel_err(mu, g["mu"]) < 1e-8
    assert rel_err(Sig, g["Sig"]) < 1e-5
    assert rel_err(out["cost_var"].cpu().numpy(), g["cost_var"]) < 1e-5
    assert rel_err(out["J"].cpu().numpy(), g["J"]) < 1e-7


def test_config5_full_horizon_against_oracle_fixture(engine):
    """BASELINE configs[4] at its REAL horizon: N = 4096, D = 16, A = 4, H = 50, one candidate, vs
    tests/golden/oracle_c5_h50.npz (CPU oracle, `tools/gen_golden_c5.py --steps 50 --candidates 1`, 39 minutes on 8 cores;
    the reference formulation cannot run this size).  VERDICT r2: the H = 50 recurrence at D = 16 was extrapolated from 5
    steps; here every one of the 50 steps is compared and the per-step errors are recorded."""
    g = load("oracle_c5_h50")
    H = int(g["H"])
    w = synth.make_workload(int(g["N"]), int(g["D"]), int(g["A"]), H, int(g["B"]), seed=int(g["seed"]))
    assert np.allclose([w.X.sum(), w.Y.sum(), w.actions.sum()], g["x_checksum"], rtol=0, atol=1e-9)   # same inputs
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    mu, Sig = out["mu"].cpu().numpy(), out["Sig"].cpu().numpy()
    per_step = [rel_err(Sig[:, t], g["Sig"][:, t]) for t in range(1, H + 1)]
    per_step_mu = [rel_err(mu[:, t], g["mu"][:, t]) for t in range(1, H + 1)]
    record("config5_full_horizon", mu=rel_err(mu, g["mu"]), Sig=rel_err(Sig, g["Sig"]), J=rel_err(out["J"].cpu().numpy(), g["J"]),
           Sig_worst_step=float(np.argmax(per_step) + 1), Sig_step1=per_step[0], Sig_step10=per_step[9], Sig_step25=per_step[24],
           Sig_step50=per_step[49], mu_step50=per_step_mu[49])
    assert max(per_step_mu) < 1e-8
    assert max(per_step) < 1e-5
    assert rel_err(out["cost_var"].cpu().numpy(), g["cost_var"]) < 1e-5
    assert rel_err(out["J"].cpu().numpy(), g["J"]) < 1e-7


def test_config5_in_range_horizon_against_oracle_fixture(engine):
    """Config 5's size with a state that STAYS where the memory is: N = 4096, D = 16, A = 4, H = 20, four
    candidates, targets that pull every state towards 0.5 and a DENSE initial covariance (synth `dynamics="contracting"`,
    `dense_s0`), vs tests/golden/oracle_c5_inrange.npz (CPU oracle, `tools/gen_golden_c5.py --inrange --steps 20 --candidates 4`,
    ~90 minutes).  In the H = 50 fixture the mean leaves [0, 1] after ~10 steps and the GP falls back to its prior (Sigma diag
    grows to 1.6); here Sigma diag runs 1.7e-2 -> 5e-4 .. 2e-3 and every step exercises the data-dependent terms.  Every step
    is compared relative to THAT step's covariance scale."""
    g = load("oracle_c5_inrange")
    H, B = int(g["H"]), int(g["B"])
    w = synth.make_workload(int(g["N"]), int(g["D"]), int(g["A"]), H, B, seed=int(g["seed"]), dynamics="contracting", dense_s0=0.02)
    assert np.allclose([w.X.sum(), w.Y.sum(), w.actions.sum()], g["x_checksum"], rtol=0, atol=1e-9)   # same inputs
    assert np.abs(w.S0 - np.diag(np.diag(w.S0))).max() > 1e-3                                            # dense Sigma_0
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    mu, Sig = out["mu"].cpu().numpy(), out["Sig"].cpu().numpy()
    assert g["mu"].min() > 0.0 and g["mu"].max() < 1.0                                                   # in range over the whole horizon
    per_step = [rel_err(Sig[:, t], g["Sig"][:, t]) for t in range(1, H + 1)]
    per_step_mu = [rel_err(mu[:, t], g["mu"][:, t]) for t in range(1, H + 1)]
    record("config5_in_range_horizon", mu=max(per_step_mu), Sig=max(per_step), J=rel_err(out["J"].cpu().numpy(), g["J"]),
           Sig_worst_step=float(np.argmax(per_step) + 1), Sig_step1=per_step[0], Sig_step5=per_step[4], Sig_step10=per_step[9],
           Sig_step20=per_step[19], Sig_diag_max_step20=float(np.diagonal(g["Sig"][:, H], axis1=-1, axis2=-2).max()))
    assert max(per_step_mu) < 1e-8
    assert max(per_step) < 1e-5
    assert rel_err(out["cost_var"].cpu().numpy(), g["cost_var"]) < 1e-5
    assert rel_err(out["J"].cpu().numpy(), g["J"]) < 1e-7


def test_random_shapes_against_oracle(engine):
    """Seeded fuzz over shapes (N not a multiple of 4 / 16 / 64, single points, padded D, time input, small and
    large input variance): layout, padding and chunking edge cases of both kernels."""
    rng = np.random.default_rng(2024)
    for case in range(24):
        N = int(rng.choice([1, 2, 3, 5, 17, 31, 63, 64, 65, 97, 130, 199, 257]))
        D = int(rng.integers(1, 7))
        A = int(rng.integers(1, 4))
        H = int(rng.integers(1, 5))
        B = int(rng.integers(1, 9))
        tm = bool(rng.integers(0, 2))
        s0 = float(rng.choice([1e-6, 1e-3, 5e-2]))
        stream = bool(rng.integers(0, 2))
        w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=1000 + case, s0=s0, noise_var=1e-4,
                                time0=float(N) if tm else 0.0)
        f = factors_of(w)
        ref = orc.evaluate_candidates(f, w)
        engine.set_option("force_global_scratch", int(stream))
        try:
            engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
            _set_cost(engine, w)
            out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        finally:
            engine.set_option("force_global_scratch", 0)
        tag = (case, N, D, A, H, B, tm, s0, stream)
        assert rel_err(out["mu"].cpu().numpy(), ref["mu"]) < 1e-8, tag
        dS = np.max(np.abs(out["Sig"].cpu().numpy() - ref["Sig"]))
        assert dS < 1e-6 * np.max(np.abs(ref["Sig"])) + 2e-11, (tag, dS)
        assert rel_err(out["J"].cpu().numpy(), ref["J"]) < 1e-7, tag


def test_rollout_is_bitwise_reproducible(engine):
    w = synth.make_workload(120, 3, 1, 10, 64, seed=5)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    a = engine.rollout(w.actions, w.mu0, w.S0)
    b = engine.rollout(w.actions, w.mu0, w.S0)
    for k in a:
        assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()), k


def test_candidates_are_independent(engine):
    """A candidate's result does not depend on what else is in the batch (sharding property)."""
    w = synth.make_workload(100, 3, 1, 8, 40, seed=6)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    full = engine.rollout(w.actions, w.mu0, w.S0)["J"].cpu().numpy()
    part = engine.rollout(w.actions[13:29], w.mu0, w.S0)["J"].cpu().numpy()
    assert np.array_equal(full[13:29], part)


def test_results_do_not_depend_on_the_workgroup_size():
    """Small memories with large batches run with narrower workgroups (rollout.hip: 512 / 256 threads from B = 2 / 8 per CU at
    N <= 64).  The sums are formed per work item in a fixed order, so the results are the same bits for every workgroup size
    -- and the batch-independence (sharding) property survives the batch-dependent choice."""
    import gp_mpc_amd
    w = synth.make_workload(50, 3, 1, 15, 600, seed=6)
    eng = gp_mpc_amd.HipEngine(0)
    try:
        eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        _set_cost(eng, w)
        res = {}
        for nt in (1024, 512, 256):
            eng.set_option("threads", nt)
            res[nt] = {k: v.cpu().numpy().copy() for k, v in eng.rollout(w.actions, w.mu0, w.S0).items()}
        for nt in (512, 256):
            for k in ("J", "mu", "Sig"):
                assert np.array_equal(res[nt][k], res[1024][k]), (nt, k)
        eng.set_option("threads", 0)
        full = eng.rollout(w.actions, w.mu0, w.S0)["J"].cpu().numpy()            # 600 candidates: 512 threads
        part = eng.rollout(w.actions[13:29], w.mu0, w.S0)["J"].cpu().numpy()     # 16 candidates: 1024 threads
        assert np.array_equal(full, res[1024]["J"])
        assert np.array_equal(full[13:29], part)
    finally:
        eng.close()


def test_argmin_trace_matches_reference(engine):
    g = load("argmin_trace")
    w = workload_of(g)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(g["cand_actions"], w.mu0, w.S0)
    assert rel_err(out["J"].cpu().numpy(), g["cand_J"]) < 1e-7
    best_J, best = engine.argmin(out["J"])
    assert np.array_equal(g["cand_actions"][best], g["best_actions"])
    assert best_J == out["J"].cpu().numpy()[best]


def test_device_side_winner_selection(engine):
    """gpmpc_argmin_async + packed record == the synchronous argmin (single rank)."""
    import torch
    from gp_mpc_amd import sharding
    rng = np.random.default_rng(3)
    J = rng.uniform(size=300)
    J[[41, 250]] = J.min() - 1.0
    acts = torch.as_tensor(rng.uniform(size=(300, 5, 2)), device=engine.device)
    bJ, bi, win = sharding.select_best_on_device(engine, torch.as_tensor(J, device=engine.device), acts, lo=1000, num_candidates=5000)
    assert bi == 1041 and bJ == J[41] and np.array_equal(win.numpy(), acts[41].cpu().numpy())
    assert engine.argmin(J, first_global_index=1000) == (J[41], 1041)
    Jn = J.copy(); Jn[0] = np.nan
    assert sharding.select_best_on_device(engine, torch.as_tensor(Jn, device=engine.device), acts, 0, 300)[1] == 0
    assert sharding.select_best_on_device(engine, torch.as_tensor(Jn, device=engine.device), acts, 7, 300)[1] == 48


def test_asynchronous_winner_selection_matches_the_blocking_one(engine):
    """sharding.select_best_async (pinned buffer + event, what bench.py pipelines) vs select_best_on_device."""
    from gp_mpc_amd import sharding
    import torch
    w = synth.make_workload(60, 3, 1, 6, 40, seed=12)
    f = factors_of(w)
    engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    _set_cost(engine, w)
    acts = torch.as_tensor(w.actions, device=engine.device)
    out = engine.rollout(acts, w.mu0, w.S0)
    J0, i0, a0 = sharding.select_best_on_device(engine, out["J"], acts, 0, 40)
    pend = sharding.select_best_async(engine, out["J"], acts, 0, 40)
    again = sharding.select_best_async(engine, out["J"], acts, 0, 40, host_buffer=None, record=pend.record)
    J1, i1, a1 = pend.result()
    J2, i2, a2 = again.result()
    assert (J0, i0) == (J1, i1) == (J2, i2) and torch.equal(a0, a1) and torch.equal(a0, a2)
    assert i0 == int(np.argmin(out["J"].cpu().numpy()))


def test_argmin_rule(engine):
    import torch
    cases = [([3.0, 1.0, 1.0, 2.0], 1), ([float("nan"), 1.0], 0), ([2.0, float("nan"), 1.0], 2),
             ([5.0], 0)]
    for vals, want in cases:
        _, idx = engine.argmin(torch.tensor(vals, dtype=torch.float64))
        assert idx == want == orc.first_wins_argmin(np.array(vals))
    big = np.random.default_rng(0).uniform(size=5000)
    big[[77, 4000]] = -1.0
    assert engine.argmin(big)[1] == 77


def test_full_size_c2_against_oracle_subset(engine):
    """BASELINE configs[1] at full size (N=200, H=25, B=256); oracle on a candidate subset."""
    w = synth.named("c2")
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    f = factors_of(w)
    sub = [0, 1, 100, 255]
    ref = orc.evaluate_candidates(f, w, actions=w.actions[sub])
    assert rel_err(out["mu"].cpu().numpy()[sub], ref["mu"]) < 1e-8
    assert rel_err(out["Sig"].cpu().numpy()[sub], ref["Sig"]) < 2e-6
    assert rel_err(out["J"].cpu().numpy()[sub], ref["J"]) < 1e-7
    Sig = out["Sig"].cpu().numpy()
    assert np.max(np.abs(Sig - Sig.transpose(0, 1, 3, 2))) == 0.0      # built symmetric
    assert np.isfinite(out["J"].cpu().numpy()).all()


@pytest.mark.parametrize("N,D,A,tm", [(40, 3, 1, False), (200, 3, 1, False), (130, 4, 2, True), (65, 1, 1, False), (300, 6, 3, False),
                                      (1100, 2, 1, False)])
def test_training_loss_and_gradient(N, D, A, tm):
    """gpmpc_mll: -log p(y | X, theta) / N of every GP and its gradient wrt (lengthscales, outputscale, noise), the loss
    of the reference's LBFGS training loop (gp_model.py:262-275), vs the closed form of oracle/gp_training.py.
    fp64: 1e-9 on the loss; the gradient goes through the explicit inverse (cond K ~ 1e5..1e6): 1e-7."""
    import gp_mpc_amd
    from oracle import gp_training
    w = synth.make_workload(N, D, A, 3, 2, include_time=tm, seed=N)
    e = gp_mpc_amd.HipEngine(0)
    try:
        out = e.mll(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        again = e.mll(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        for k in out:
            assert np.array_equal(out[k], again[k])                  # fixed-order sums: bitwise reproducible
        for a in range(D):
            loss, g_ls, g_os, g_nz = gp_training.neg_mll_and_grad(w.X, w.Y[:, a], w.lengthscales[a], w.outputscales[a], w.noises[a])
            assert abs(out["loss"][a] - loss) < 1e-9 * abs(loss)
            assert rel_err(out["d_lengthscale"][a], g_ls) < 1e-7
            assert abs(out["d_outputscale"][a] - g_os) < 1e-7 * abs(g_os)
            assert abs(out["d_noise"][a] - g_nz) < 1e-7 * abs(g_nz)
        # the factors left behind are those of these hyper-parameters
        iK, beta = e.factors()
        iK0, beta0 = orc.factorize(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        assert rel_err(beta.cpu().numpy(), beta0) < 1e-8
    finally:
        e.close()
