"""Tier 2 (GPU, through the C ABI): the batch-major path of the rollout (csrc/pair_tile_kernel.h, point_pass_kernel.h) --
per horizon step the N x N work of the diagonal output pairs by workgroups that own a tile of beta beta^T - iK and loop
over the candidates, the O(N) rest per candidate -- against the reference's goldens, the CPU oracle and the fused-horizon
kernel, and the full-size batches of BASELINE configs[2] / configs[3].

Covariance tolerances follow tests/test_gpu_parity.py: two correct fp64 evaluations of the moment-matched step differ by the
method's own noise floor (cancellation of N^2 terms), so the batch-major path is required to be as close to the checker
as the fused kernel is, and close to the fused kernel itself."""
import numpy as np
import pytest

from helpers import load, workload_of, factors_of, rel_err, record
from oracle import gpmpc_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu

FUSED, STREAM, TILES = 0, 1, 2


@pytest.fixture(scope="module")
def engine():
    import gp_mpc_amd
    eng = gp_mpc_amd.HipEngine(0)
    yield eng
    eng.close()


def _rollout(engine, w, tiles, **options):
    """tiles: 1 = force the batch-major path, 2 = forbid it; options are reset afterwards."""
    engine.set_option("pair_tiles", tiles)
    for k, v in options.items():
        engine.set_option(k, v)
    try:
        out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
        path = engine.last_rollout_path
    finally:
        for k in options:
            engine.set_option(k, 0)
        engine.set_option("pair_tiles", 0)
    return {k: v.cpu().numpy() for k, v in out.items()}, path


@pytest.mark.parametrize("name", ["traj_c4", "traj_c4_n1000", "traj_c4_time", "traj_c2", "traj_c3", "traj_c1", "traj_clip", "traj_constraints",
                                  "traj_bigvar"])
def test_reference_goldens_through_the_batch_major_path(engine, name):
    """The reference's own trajectories (gp_model.py:60-180 + reward mapper + LCB objective) with the path forced."""
    g = load(name)
    w = workload_of(g)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    clip = bool(g["clip"]) if "clip" in g else False
    smin = g["state_min"] if "use_constraints" in g and bool(g["use_constraints"]) else None
    engine.set_cost(w.target, w.W, w.W_T, w.kappa, clip, smin, g["state_max"] if smin is not None else None)
    out, path = _rollout(engine, w, 1)
    assert path == TILES
    tol = {"traj_c3": 1e-5, "traj_c2": 1e-6, "traj_c4": 1e-6, "traj_c4_n1000": 1e-6}.get(name, 1e-7)      # as tests/test_gpu_parity.py SIG_TOL
    e_mu, e_S, e_J = rel_err(out["mu"], g["mu"]), rel_err(out["Sig"], g["Sig"]), rel_err(out["J"], g["J"])
    record(f"batch_major_vs_reference[{name}]", mu=e_mu, Sig=e_S, J=e_J)
    assert e_mu < 1e-8 and e_S < tol and e_J < 1e-7
    assert rel_err(-out["cost_mu"], g["rewards"]) < 1e-8


@pytest.mark.parametrize("N,D,A,H,B,tm,s0,opts", [
    (300, 4, 2, 4, 5, False, 1e-6, {}), (129, 4, 2, 3, 3, False, 1e-4, {}), (257, 3, 1, 4, 4, False, 1e-5, {}),
    (200, 2, 1, 5, 7, False, 1e-6, {}), (140, 1, 1, 4, 3, False, 1e-5, {}), (130, 3, 1, 3, 3, True, 1e-5, {}),
    (300, 4, 2, 3, 4, False, 3e-2, {}),                      # large input variance: high Taylor degrees / direct exp, element-wise hand-over
    (150, 4, 2, 3, 70, False, 1e-5, {}),                     # more candidates than one tile chunk
    (260, 4, 2, 3, 4, False, 1e-5, {"force_path": 1}),       # direct exp everywhere: every candidate goes to the element-wise kernel
    (260, 4, 2, 3, 4, False, 1e-5, {"force_path": 2}),       # Taylor, never separable: same hand-over, Taylor tiles
    (128, 2, 2, 3, 2, False, 1e-5, {}), (40, 3, 1, 3, 3, False, 1e-5, {}), (1, 2, 1, 2, 2, False, 1e-5, {}),
    (200, 3, 1, 3, 5, False, 2e-3, {}), (200, 2, 1, 3, 5, False, 5e-3, {}),          # degrees beyond one band of monomials
])
def test_batch_major_path_against_oracle_and_fused_kernel(engine, N, D, A, H, B, tm, s0, opts):
    w = synth.make_workload(N, D, A, H, B, include_time=tm, seed=N + D, s0=s0, time0=3.0 if tm else 0.0)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    t_out, p_t = _rollout(engine, w, 1, **opts)
    f_out, p_f = _rollout(engine, w, 2, **opts)
    assert (p_t, p_f) == (TILES, FUSED)
    ref = orc.evaluate_candidates(factors_of(w), w)
    fused_vs_oracle = rel_err(f_out["Sig"], ref["Sig"])
    e = dict(mu=rel_err(t_out["mu"], ref["mu"]), Sig=rel_err(t_out["Sig"], ref["Sig"]), J=rel_err(t_out["J"], ref["J"]),
             Sig_vs_fused=rel_err(t_out["Sig"], f_out["Sig"]), fused_Sig=fused_vs_oracle)
    record(f"batch_major_vs_oracle[N{N},D{D},A{A},H{H},B{B},t{int(tm)},s{s0:g},{sorted(opts.items())}]", **e)
    assert e["mu"] < 1e-8
    assert e["Sig"] < max(2e-6, 2.0 * fused_vs_oracle)
    assert e["Sig_vs_fused"] < max(2e-6, fused_vs_oracle)
    assert e["J"] < 1e-6
    assert np.max(np.abs(t_out["Sig"] - t_out["Sig"].transpose(0, 1, 3, 2))) == 0.0


def test_batch_major_path_is_reproducible_and_independent_of_the_batch(engine):
    w = synth.make_workload(300, 4, 2, 3, 40, seed=5, s0=1e-5)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    full, _ = _rollout(engine, w, 1)
    again, _ = _rollout(engine, w, 1)
    assert np.array_equal(full["Sig"], again["Sig"]) and np.array_equal(full["J"], again["J"])
    w.actions = w.actions[3:20].copy()
    sub, _ = _rollout(engine, w, 1)
    assert np.array_equal(full["Sig"][3:20], sub["Sig"]) and np.array_equal(full["mu"][3:20], sub["mu"])
    assert np.array_equal(full["J"][3:20], sub["J"])
    other_chunk, _ = _rollout(engine, w, 1, tile_chunk=6)           # another split of the candidates over workgroups
    assert np.array_equal(other_chunk["Sig"], sub["Sig"])


def test_mixed_batch_hands_some_candidates_to_the_element_wise_kernel(engine):
    """A candidate whose predicted covariance leaves the separable range at SOME step (Taylor degree beyond the monomial
    table, or the direct-exp form) is handed to the element-wise kernel for that step only.  All candidates share
    (mu0, S0), so a mix arises from the actions: a starting variance near the edge of the range, covariances that then
    grow differently along the horizon."""
    w = synth.make_workload(180, 4, 2, 8, 12, seed=11, s0=4e-3)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    t_out, p_t = _rollout(engine, w, 1)
    f_out, _ = _rollout(engine, w, 2)
    ref = orc.evaluate_candidates(factors_of(w), w)
    assert p_t == TILES
    assert rel_err(t_out["mu"], ref["mu"]) < 1e-8
    assert rel_err(t_out["Sig"], ref["Sig"]) < max(2e-6, 2.0 * rel_err(f_out["Sig"], ref["Sig"]))
    assert rel_err(t_out["J"], ref["J"]) < 1e-6


def test_gradient_uses_the_batch_major_forward_pass(engine):
    """gpmpc_rollout_grad's forward rollout goes through the same dispatch; J and dJ/du must not depend on the path."""
    w = synth.make_workload(200, 4, 2, 4, 6, seed=2, s0=1e-5)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    res = {}
    for tiles in (1, 2):
        engine.set_option("pair_tiles", tiles)
        try:
            g = engine.rollout_grad(w.actions, w.mu0, w.S0)
            res[tiles] = (g["J"].cpu().numpy(), g["grad"].cpu().numpy(), engine.last_rollout_path)
        finally:
            engine.set_option("pair_tiles", 0)
    assert res[1][2] == TILES and res[2][2] == FUSED
    assert rel_err(res[1][0], res[2][0]) < 1e-9
    assert rel_err(res[1][1], res[2][1]) < 1e-6


# ------------------------------------------------------------------------------------------ full-size BASELINE batches
def _full_size(engine, name, sub, path, sig_tol):
    w = synth.named(name)
    N, D, A, E, H, B = w.dims
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    out = engine.rollout(w.actions, w.mu0, w.S0)
    assert engine.last_rollout_path == path
    mu, Sig, J = out["mu"].cpu().numpy(), out["Sig"].cpu().numpy(), out["J"].cpu().numpy()
    ref = orc.evaluate_candidates(factors_of(w), w, actions=w.actions[sub])
    e_mu, e_S, e_J = rel_err(mu[sub], ref["mu"]), rel_err(Sig[sub], ref["Sig"]), rel_err(J[sub], ref["J"])
    record(f"full_size[{name}]", mu=e_mu, Sig=e_S, J=e_J)
    assert e_mu < 1e-8 and e_S < sig_tol and e_J < 1e-6
    assert np.isfinite(J).all() and np.isfinite(Sig).all()
    assert np.max(np.abs(Sig - Sig.transpose(0, 1, 3, 2))) == 0.0           # built symmetric
    # the same candidates as a batch of their own, same path: bit for bit
    lo, hi = B // 2 - 8, B // 2 + 56
    w.actions = w.actions[lo:hi].copy()
    engine.set_option("pair_tiles", 1 if path == TILES else 2)
    engine.set_option("cluster", 1)          # "same path": 64 candidates alone would take the few-candidate cooperative form (its own chunk length)
    try:
        part = engine.rollout(w.actions, w.mu0, w.S0)
        assert engine.last_rollout_path == path
    finally:
        engine.set_option("pair_tiles", 0)
        engine.set_option("cluster", 0)
    assert np.array_equal(part["Sig"].cpu().numpy(), Sig[lo:hi]) and np.array_equal(part["J"].cpu().numpy(), J[lo:hi])


def test_full_size_c3_batch(engine):
    """BASELINE configs[2] at full size (N = 500, D = 2, H = 40, B = 1024): the fused-horizon kernel (tables L2-resident)."""
    _full_size(engine, "c3", [0, 1, 511, 1023], FUSED, 1e-5)        # vs the numpy oracle (itself 8e-6 from the exact values): 5.3e-6 achieved


def test_full_size_c4_batch(engine):
    """BASELINE configs[3] at full size (N = 1000, D = 4, H = 30, B = 2048): the batch-major path is what the dispatch picks."""
    _full_size(engine, "c4", [0, 1, 1024, 2047], TILES, 5e-6)
