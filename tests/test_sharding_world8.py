"""Tier 3 (CPU, gloo): the sharded searches at the world size the node has (8 ranks), with B < world and B % world != 0, and the
failure / disagreement handling of the sharded paths:

  * a failure on ONE rank must not leave the others waiting in a collective: every rank raises;
  * sharding is opt-in (`ControllerConfig.shard_over_ranks`): a process group initialised for another purpose changes nothing;
  * ranks that seed numpy differently still search the single-GPU population (the draws are rank 0's);
  * ranks at different states are detected inside the exchange.

The evaluator is the CPU stand-in of tests/stub_engine.py (the HIP engine needs a GPU); what is under test is everything around
the launches."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _spawn(fn, world, port_base, *args, timeout=240):
    """Run fn(rank, world, port, ret, *args) on `world` gloo ranks; a rank still alive after `timeout` s is a deadlock."""
    port = port_base + (os.getpid() * 7 + world * 31 + sum(len(str(a)) for a in args)) % 1500
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_entry, args=(world, port, ret, fn.__name__, args), nprocs=world, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=1.0):
        if time.time() - t0 > timeout:
            for p in ctx.processes:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f"{fn.__name__}: ranks still running after {timeout} s (deadlock); finished: {sorted(ret.keys())}")
    return dict(ret)


def _entry(rank, world, port, ret, name, args):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[name](rank, world, *args)
    except Exception as e:          # noqa: BLE001 -- the tests look at what each rank raised
        ret[rank] = ("raised", type(e).__name__, str(e)[:200])
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ device cross-entropy search
def _cem_setup():
    from oracle import synth
    from stub_engine import OracleEngine
    w = synth.make_workload(N=24, D=2, A=1, H=3, B=1, seed=4)
    eng = OracleEngine()
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    return w, eng


def _cem_case(B, iters, n_elite, group, fail_rank=-1, rank=0):
    from gp_mpc_amd import sharding
    w, eng = _cem_setup()
    rng = np.random.default_rng(12)
    noise = np.concatenate([rng.uniform(size=(1, B, 3)), rng.standard_normal((iters - 1, B, 3))])
    if rank == fail_rank:
        real = eng.cem_local

        def failing(*a, **k):
            if a[7] >= 1:                    # iteration 1: after one healthy round of collectives
                raise RuntimeError("injected failure of this rank's slice")
            return real(*a, **k)
        eng.cem_local = failing
    x, J = sharding.sharded_cem_search(eng, w.mu0, w.S0, B, 3, 1, iters, n_elite, seed=0, noise=noise, group=group)
    return x, J, eng.launches


def cem_rank(rank, world, B, n_elite, fail_rank):
    return _cem_case(B, 3, n_elite, None, fail_rank, rank)


@pytest.mark.parametrize("B,n_elite", [(5, 2), (13, 4)])
def test_cross_entropy_search_at_world_8(B, n_elite):
    """B < world (three ranks with an empty slice) and B % world != 0: every rank ends with the single-rank winner, bitwise."""
    from gp_mpc_amd import sharding
    want = _cem_case(B, 3, n_elite, sharding.LOCAL)
    got = _spawn(cem_rank, 8, 41000, B, n_elite, -1)
    for r in range(8):
        x, J, launches = got[r]
        assert np.array_equal(x, want[0]) and J == want[1], (r, got[r], want)
        lo, hi = sharding.shard_bounds(B, 8, r)
        assert (launches > 0) == (hi > lo)


def test_cross_entropy_search_failure_on_one_rank_raises_everywhere():
    """Rank 1's cem_local fails in iteration 1: it keeps entering the all_gathers with marked records, and every rank raises
    after the last one instead of waiting in a collective."""
    got = _spawn(cem_rank, 3, 42500, 9, 2, 1, timeout=120)
    assert got[1][0] == "raised" and "injected" in got[1][2], got[1]
    for r in (0, 2):
        assert got[r][0] == "raised" and "another rank" in got[r][2], (r, got[r])


def cem_merge_fail_rank(rank, world, B, n_elite, fail_rank):
    """The LAST iteration's merge fails on one rank: no later gather could carry a marker."""
    from gp_mpc_amd import sharding
    w, eng = _cem_setup()
    rng = np.random.default_rng(12)
    iters = 2
    noise = np.concatenate([rng.uniform(size=(1, B, 3)), rng.standard_normal((iters - 1, B, 3))])
    if rank == fail_rank:
        real = eng.cem_merge

        def failing(elites, n_elite_, n, it, state):
            if it == iters - 1:
                raise RuntimeError("injected failure of this rank's last merge")
            return real(elites, n_elite_, n, it, state)
        eng.cem_merge = failing
    x, J = sharding.sharded_cem_search(eng, w.mu0, w.S0, B, 3, 1, iters, n_elite, seed=0, noise=noise, group=None)
    return x, J


def test_cross_entropy_search_failure_of_the_last_merge_raises_everywhere():
    """a cem_merge failure in the LAST iteration is followed by no gather; the ranks agree on it through one
    all_reduce of their failure flags, so nobody returns a winner while another rank raises."""
    got = _spawn(cem_merge_fail_rank, 3, 42900, 9, 2, 2, timeout=120)
    assert got[2][0] == "raised" and "injected" in got[2][2], got[2]
    for r in (0, 1):
        assert got[r][0] == "raised" and "another rank" in got[r][2], (r, got[r])


def test_cross_entropy_search_beyond_the_merge_limit_runs_unsharded():
    """world x n_elite > 4096 records do not fit the merge kernel's LDS sort: every rank runs the whole population (same draws),
    no collective, instead of failing with GPMPC_ERR_LIMIT only when sharded."""
    from gp_mpc_amd import sharding
    calls = {}

    class Eng:
        device = torch.device("cpu")

        def cem_search(self, *a, **k):
            calls["search"] = (a[2], k["seed"])
            return np.zeros(3), 1.5

        def cem_local(self, *a, **k):
            raise AssertionError("must not shard")
    old = (dist.is_available, dist.is_initialized, dist.get_world_size, dist.get_rank)
    try:
        sharding.dist.is_initialized = lambda: True
        sharding.dist.get_world_size = lambda group=None: 8
        sharding.dist.get_rank = lambda group=None: 3
        x, J = sharding.sharded_cem_search(Eng(), np.zeros(2), np.eye(2), 4096, 3, 1, 2, 600, seed=11)
    finally:
        dist.is_available, dist.is_initialized, dist.get_world_size, dist.get_rank = old
    assert calls["search"] == (4096, 11) and J == 1.5


# ------------------------------------------------------------------------------------------ controller paths
_LBFGS = {"disp": None, "maxcor": 4, "ftol": 1e-15, "gtol": 1e-15, "eps": 1e-2, "maxfun": 3, "maxiter": 3, "iprint": -1, "maxls": 4,
          "finite_diff_rel_step": None}


def _controller(kind, restarts, shard=True, mu_shift=0.0, fail=False):
    from helpers import make_controller
    from oracle import synth
    from stub_engine import OracleEngine
    w = synth.make_workload(N=20, D=2, A=1, H=3, B=1, seed=6)
    eng = OracleEngine()
    if fail:
        def broken(*a, **k):
            raise RuntimeError("injected failure of this rank's launch")
        eng.rollout_grad = broken
    c = make_controller(w, optimize=(kind == "lbfgs"), restarts=restarts, engine=eng, optimizer_params=_LBFGS, shard=shard)
    if kind == "lbfgs":
        c.config.controller.candidate_optimizer = "lbfgs"
        c.config.controller.lbfgs_candidates = restarts
    return w, eng, c, w.mu0 + mu_shift


def _steps(c, mu0, eng, n=2):
    out = []
    for _ in range(n):
        a = c.get_action(mu0)
        out.append(dict(action=np.asarray(a), best=int(c.best_candidate_index), J=float(c.best_candidate_J),
                        prev=c.actions_mpc_previous_iter.copy(), states=np.asarray(c.get_iter_info().predicted_states),
                        launches=eng.launches))
    return out


def controller_rank(rank, world, kind, restarts, seed_by_rank, shard, shift_rank, fail_rank):
    w, eng, c, mu0 = _controller(kind, restarts, shard=shard, mu_shift=0.01 if rank == shift_rank else 0.0, fail=(rank == fail_rank))
    np.random.seed(8 + (rank if seed_by_rank else 0))
    return _steps(c, mu0, eng)


def _single(kind, restarts):
    w, eng, c, mu0 = _controller(kind, restarts)
    np.random.seed(8)
    return _steps(c, mu0, eng)


def _same(got, want, keys=("action", "prev", "states")):
    for step, (g, e) in enumerate(zip(got, want)):
        assert g["best"] == e["best"] and g["J"] == e["J"], (step, g["best"], e["best"], g["J"], e["J"])
        for k in keys:
            assert np.array_equal(g[k], e[k]), (step, k)


@pytest.mark.parametrize("kind,restarts", [("shoot", 5), ("shoot", 13), ("lbfgs", 3), ("lbfgs", 11)])
def test_controller_searches_at_world_8(kind, restarts):
    """Random shooting and the lockstep L-BFGS restarts over 8 ranks, fewer candidates than ranks and a ragged split: every rank
    returns the single-process action, winner and logging caches."""
    want = _single(kind, restarts)
    got = _spawn(controller_rank, 8, 43500, kind, restarts, False, True, -1, -1)
    for r in range(8):
        assert isinstance(got[r], list), (r, got[r])
        _same(got[r], want)


def test_lbfgs_restarts_do_not_depend_on_the_ranks_seeds():
    """Ranks that seed numpy differently: the starting points are rank 0's (one broadcast per control step), so the union of the
    slices is still the single-process set of restarts -- the FIRST control step matches the single-process run seeded like rank 0
    (later steps draw from generators that have diverged, by construction of the test)."""
    want = _single("lbfgs", 5)
    got = _spawn(controller_rank, 3, 44500, "lbfgs", 5, True, True, -1, -1)
    for r in range(3):
        assert isinstance(got[r], list), (r, got[r])
        _same(got[r][:1], want[:1])
        for k in ("action", "prev", "states"):       # and the ranks agree with each other at every step
            assert np.array_equal(got[r][1][k], got[0][1][k])


def test_lbfgs_failure_on_one_rank_raises_everywhere():
    got = _spawn(controller_rank, 3, 45500, "lbfgs", 6, False, True, -1, 1, timeout=120)
    assert got[1][0] == "raised" and ("injected" in got[1][2] or "batched evaluation failed" in got[1][2]), got[1]
    for r in (0, 2):
        assert got[r][0] == "raised" and "rank(s) [1]" in got[r][2], (r, got[r])


def shoot_fail_rank(rank, world, restarts, fail_rank):
    w, eng, c, mu0 = _controller("shoot", restarts)
    if rank == fail_rank:
        def broken(*a, **k):
            raise RuntimeError("injected failure of this rank's launch")
        eng.rollout = broken
    np.random.seed(8)
    return _steps(c, mu0, eng, n=1)


@pytest.mark.parametrize("restarts,fail_rank", [(7, 1), (2, 0)])
def test_random_shooting_failure_on_one_rank_raises_everywhere(restarts, fail_rank):
    """a slice evaluation that raises on one rank (launch error, GPMPC_ERR_LIMIT) used to happen BEFORE the
    step's all_gather -- the peers then waited in it until the watchdog fired.  The failing rank now contributes an (inf, -1)
    record with an error flag and every rank raises after the exchange (also when the failing rank owned the only candidates
    of interest: B = 2 over 3 ranks, rank 0 failing)."""
    got = _spawn(shoot_fail_rank, 3, 45900, restarts, fail_rank, timeout=120)
    assert got[fail_rank][0] == "raised" and "injected" in got[fail_rank][2], got[fail_rank]
    for r in range(3):
        if r != fail_rank:
            assert got[r][0] == "raised" and f"rank(s) [{fail_rank}]" in got[r][2], (r, got[r])


def cem_device_state_rank(rank, world, shift_rank):
    w, eng, c, mu0 = _controller("shoot", 4)
    c.config.controller.optimize = True
    c.config.controller.candidate_optimizer = "cem_device"
    c.config.controller.cem_candidates, c.config.controller.cem_iterations, c.config.controller.cem_elite_fraction = 6, 2, 0.34
    np.random.seed(8)
    return _steps(c, mu0 + (0.01 if rank == shift_rank else 0.0), eng, n=1)


def test_device_cross_entropy_search_detects_ranks_at_different_states():
    """the sharded device CEM adopted rank 0's state silently; a rank called with another observation now makes
    every rank raise (one all_reduce of the mismatch flags right after the broadcast)."""
    got = _spawn(cem_device_state_rank, 2, 46900, 1, timeout=120)
    for r in range(2):
        assert got[r][0] == "raised" and "disagree on the state" in got[r][2], (r, got[r])


@pytest.mark.parametrize("kind", ["shoot", "lbfgs"])
def test_ranks_at_different_states_are_detected(kind):
    got = _spawn(controller_rank, 2, 46500, kind, 4, False, True, 1, -1, timeout=120)
    for r in range(2):
        assert got[r][0] == "raised" and "disagree on the state" in got[r][2], (r, got[r])


@pytest.mark.parametrize("kind", ["shoot", "lbfgs"])
def test_sharding_is_opt_in(kind):
    """A process group initialised for some other purpose, `shard_over_ranks` left at its default: every rank evaluates ALL its
    candidates itself (no collective; ranks at different states do not mix slices)."""
    want = _single(kind, 4)
    got = _spawn(controller_rank, 2, 47500, kind, 4, False, False, -1, -1, timeout=120)
    for r in range(2):
        assert isinstance(got[r], list), (r, got[r])
        _same(got[r], want)
        assert got[r][0]["launches"] == want[0]["launches"]
    from gp_mpc_amd.config_classes import ControllerConfig
    assert ControllerConfig().shard_over_ranks is False
