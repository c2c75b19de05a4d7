"""Tier 3 (CPU, gloo, world_size 2 and 3): the N > 1 candidate-sharding path.  The local
evaluator is the CPU oracle here (the HIP engine needs a GPU); what is under test is the slice
arithmetic, the 16-byte gather, the keep-the-best rule across ranks and the winner broadcast."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gp_mpc_amd import sharding
        from oracle import synth
        from oracle import gpmpc_oracle as orc
        w = synth.make_workload(N=40, D=3, A=1, H=5, B=11, seed=9)
        f = orc.Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
        full = orc.evaluate_candidates(f, w)
        J_all = full["J"].copy()
        if mode == "nan0":
            J_all[0] = np.nan
        elif mode == "nan_mid":
            J_all[int(np.argmin(J_all))] = np.nan
        elif mode == "tie":
            J_all[7] = J_all[2] = J_all.min() - 1.0
        lo, hi = sharding.shard_bounds(11, world, rank)

        def evaluate(actions_local):
            J = J_all[lo:hi]
            best, val = -1, np.inf
            for k, v in enumerate(J):
                if (lo + k == 0 and np.isnan(v)):
                    best, val = 0, v
                    break
                if v < val:
                    best, val = lo + k, v
            return float(val), best
        acts = torch.as_tensor(w.actions[lo:hi])
        J, i, win = sharding.sharded_argmin(evaluate, acts, lo, 11, torch.device("cpu"))
        want = orc.first_wins_argmin(J_all)
        ok = (i == want) and np.array_equal(win.numpy(), w.actions[want])
        ok = ok and (np.isnan(J) if np.isnan(J_all[want]) else J == J_all[want])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["plain", "nan0", "nan_mid", "tie"])
def test_sharded_argmin_matches_sequential_rule(world, mode):
    port = 29500 + (os.getpid() + world * 7 + len(mode)) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, mode, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from gp_mpc_amd import sharding
    for B in [1, 7, 8, 256, 8192]:
        for world in [1, 2, 3, 8]:
            cuts = [sharding.shard_bounds(B, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
            for idx in {0, B - 1, B // 2}:
                r = sharding.owner_of(idx, B, world)
                assert cuts[r][0] <= idx < cuts[r][1]


def test_combine_best_rules():
    sys.path.insert(0, ROOT)
    from gp_mpc_amd.sharding import combine_best
    assert combine_best([(3.0, 1), (1.0, 5)]) == (1.0, 5)
    assert combine_best([(1.0, 4), (1.0, 2)]) == (1.0, 2)
    J, i = combine_best([(float("nan"), 0), (1.0, 5)])
    assert i == 0 and J != J
    assert combine_best([(float("inf"), -1), (2.0, 9)]) == (2.0, 9)
    assert combine_best([(float("inf"), -1)])[1] == -1


# ------------------------------------------------------------------- the code path that ships (missing 2)
def _shipping_worker(rank, world, port, mode, B, ret):
    """select_best_async / select_best_on_device (what bench.py and the controller call) with >1 rank."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gp_mpc_amd import sharding
        from oracle import gpmpc_oracle as orc
        from stub_engine import OracleEngine
        rng = np.random.default_rng(17)
        H, A = 4, 2
        acts_all = rng.uniform(size=(B, H, A))
        J_all = rng.uniform(size=B)
        if mode == "nan0":
            J_all[0] = np.nan
        elif mode == "nan_mid" and B > 2:
            J_all[int(np.argmin(J_all))] = np.nan
        elif mode == "tie" and B > 8:
            J_all[7] = J_all[2] = J_all.min() - 1.0
        want = orc.first_wins_argmin(J_all)
        lo, hi = sharding.shard_bounds(B, world, rank)
        eng = OracleEngine()
        acts = torch.as_tensor(acts_all[lo:hi])
        J = torch.as_tensor(J_all[lo:hi])
        ok = True
        for variant in ("blocking", "async", "async_reuse", "async_host", "extra"):
            if variant == "blocking":
                bJ, bi, win = sharding.select_best_on_device(eng, J, acts, lo, B)
            elif variant == "extra":
                payload = torch.full((5,), float(rank))
                bJ, bi, win, extras = sharding.select_best_on_device(eng, J, acts, lo, B, extra=payload)
                ok = ok and extras.shape == (world, 5) and all(float(extras[r, 0]) == r for r in range(world))
            elif variant == "async_host":
                # the records meet between the hosts when the result is read: nothing collective at selection time
                pend = sharding.select_best_async(eng, J, acts, lo, B, exchange="host")
                assert pend.host.numel() == 2 + H * A and pend.exchange_group is not None
                bJ, bi, win = pend.result()
            else:
                pend = sharding.select_best_async(eng, J, acts, lo, B)
                if variant == "async_reuse":
                    pend = sharding.select_best_async(eng, J, acts, lo, B, host_buffer=pend.host, record=pend.record)
                bJ, bi, win = pend.result()
            ok = ok and bi == want and np.array_equal(win.numpy(), acts_all[want])
            ok = ok and (np.isnan(bJ) if np.isnan(J_all[want]) else bJ == J_all[want])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 11), (3, 11), (2, 1), (3, 2)])
@pytest.mark.parametrize("mode", ["plain", "nan0", "nan_mid", "tie"])
def test_shipping_selection_paths_with_several_ranks(world, B, mode):
    """B < world leaves ranks with an EMPTY slice: they launch nothing and contribute (inf, -1)."""
    port = 31500 + (os.getpid() + world * 13 + B * 3 + len(mode)) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shipping_worker, args=(world, port, mode, B, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _controller_step(restarts, seed):
    """One get_action of a GpMpcController (random shooting) on the stub engine; returns what a caller can see."""
    from helpers import make_controller
    from oracle import synth
    from stub_engine import OracleEngine
    w = synth.make_workload(N=30, D=3, A=1, H=4, B=1, seed=3)
    eng = OracleEngine()
    c = make_controller(w, optimize=False, restarts=restarts, engine=eng)
    np.random.seed(seed)                       # the reference's global generator: every rank seeds it identically
    out = []
    for step in range(2):                      # second step exercises init_from_previous_actions
        a = c.get_action(w.mu0)
        info = c.get_iter_info()
        out.append(dict(action=np.asarray(a), best=int(c.best_candidate_index), J=float(c.best_candidate_J),
                        prev=c.actions_mpc_previous_iter.copy(),
                        states=np.asarray(info.predicted_states), std=np.asarray(info.predicted_states_std),
                        costs=np.asarray(info.predicted_costs), lcb=float(info.lower_bound_mean_predicted_cost),
                        launches=eng.launches))
    return out


def _controller_worker(rank, world, port, restarts, seed, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _controller_step(restarts, seed)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,restarts", [(2, 5), (2, 1), (3, 4)])
def test_controller_random_shooting_sharded_over_ranks(world, restarts):
    """GpMpcController.get_action with torch.distributed initialised: every rank returns the
    action, winner and logging caches (IterationInformation) of the single-process run -- including ranks that do
    not own the last candidate and ranks whose slice is empty (restarts < world)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    want = _controller_step(restarts, seed=5)
    port = 33500 + (os.getpid() + world * 17 + restarts) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_controller_worker, args=(world, port, restarts, 5, ret), nprocs=world, join=True)
    for r in range(world):
        got = ret[r]
        for step, (g, e) in enumerate(zip(got, want)):
            assert g["best"] == e["best"] and g["J"] == e["J"], (r, step)
            for k in ("action", "prev", "states", "std", "costs"):
                assert np.array_equal(g[k], e[k]), (r, step, k)
            assert g["lcb"] == e["lcb"]
    lo_hi = [(__import__("gp_mpc_amd").sharding.shard_bounds(restarts, world, r)) for r in range(world)]
    for r, (lo, hi) in enumerate(lo_hi):          # a rank with an empty slice never launched
        assert (ret[r][0]["launches"] > 0) == (hi > lo)


# ------------------------------------------------------------------- the device cross-entropy search, sharded
def _cem_run(B, iters, n_elite, first):
    from gp_mpc_amd import sharding
    from oracle import synth
    from stub_engine import OracleEngine
    w = synth.make_workload(N=30, D=2, A=1, H=3, B=1, seed=4)
    eng = OracleEngine()
    eng.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    eng.set_cost(w.target, w.W, w.W_T, w.kappa)
    rng = np.random.default_rng(12)
    noise = np.concatenate([rng.uniform(size=(1, B, 3)), rng.standard_normal((iters - 1, B, 3))])
    fc = np.full(3, 0.25) if first else None
    x, J = sharding.sharded_cem_search(eng, w.mu0, w.S0, B, 3, 1, iters, n_elite, seed=0, first_candidate=fc, noise=noise)
    return x, J, eng.launches


def _cem_worker(rank, world, port, B, iters, n_elite, first, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _cem_run(B, iters, n_elite, first)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,n_elite,first", [(2, 13, 4, False), (3, 16, 5, True), (2, 5, 4, True), (3, 2, 2, False)])
def test_sharded_cross_entropy_search_reaches_the_single_rank_state(world, B, n_elite, first):
    """Each rank draws its slice of the population from the shared key, one elite merge per iteration: every rank ends with
    exactly the winner (vector and objective, bitwise) the single-rank search reaches on the same draws -- including slices
    shorter than n_elite and empty slices (B < world)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    want = _cem_run(B, 4, n_elite, first)
    port = 35500 + (os.getpid() + world * 19 + B) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cem_worker, args=(world, port, B, 4, n_elite, first, ret), nprocs=world, join=True)
    for r in range(world):
        x, J, launches = ret[r]
        assert np.array_equal(x, want[0]) and J == want[1], (r, x, want[0], J, want[1])


# ------------------------------------------------------------------- lockstep L-BFGS restarts, sharded (missing 3)
def _lbfgs_step(restarts, seed):
    from helpers import make_controller
    from oracle import synth
    from stub_engine import OracleEngine
    w = synth.make_workload(N=25, D=2, A=1, H=3, B=1, seed=6)
    eng = OracleEngine()
    c = make_controller(w, optimize=True, restarts=restarts, engine=eng,
                        optimizer_params={"disp": None, "maxcor": 4, "ftol": 1e-15, "gtol": 1e-15, "eps": 1e-2, "maxfun": 4,
                                          "maxiter": 4, "iprint": -1, "maxls": 4, "finite_diff_rel_step": None})
    c.config.controller.candidate_optimizer = "lbfgs"
    c.config.controller.lbfgs_candidates = restarts
    np.random.seed(seed)
    out = []
    for step in range(2):
        a = c.get_action(w.mu0)
        out.append(dict(action=np.asarray(a), best=int(c.best_candidate_index), J=float(c.best_candidate_J),
                        prev=c.actions_mpc_previous_iter.copy(), states=np.asarray(c.get_iter_info().predicted_states)))
    return out


def _lbfgs_worker(rank, world, port, restarts, seed, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _lbfgs_step(restarts, seed)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,restarts", [(2, 3), (3, 2)])
def test_lockstep_lbfgs_restarts_sharded_over_ranks(world, restarts):
    """candidate_optimizer = "lbfgs" with torch.distributed initialised: each rank solves its slice of the restarts (the
    reference's loop gp_mpc_controller.py:125-141, spread), one all_gather of [fun, restart, solution]; every rank returns
    the single-process winner -- also a rank whose slice is empty (restarts < world)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    want = _lbfgs_step(restarts, seed=8)
    port = 37500 + (os.getpid() + world * 23 + restarts) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_lbfgs_worker, args=(world, port, restarts, 8, ret), nprocs=world, join=True)
    for r in range(world):
        for step, (g, e) in enumerate(zip(ret[r], want)):
            assert g["best"] == e["best"] and g["J"] == e["J"], (r, step, g["best"], e["best"], g["J"], e["J"])
            for k in ("action", "prev", "states"):
                assert np.array_equal(g[k], e[k]), (r, step, k)
