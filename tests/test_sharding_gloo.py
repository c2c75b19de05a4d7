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
