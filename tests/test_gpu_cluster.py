"""Tier 2 (GPU): the few-candidate cooperative form of the fused-horizon kernel (rollout_kernel<..., CL>, option "cluster").

The reference evaluates ONE action sequence per objective call (restarts_optim 1-2, gp_mpc_controller.py:125-141, 229-285);
with few candidates a cluster of workgroups shares each candidate's horizon step.  Checked here:
  * bit for bit the one-workgroup-per-candidate kernel at equal row-chunk length, over shapes, batch sizes, cluster sizes and
    workgroup widths (every value is formed by the same instruction sequence and summed in the same order);
  * the reference goldens through the cooperative path at its own (shorter) chunk length, same tolerances as the plain path;
  * results independent of the batch composition and of the cluster size among few-candidate launches;
  * the dispatch rule, the objective + gradient on top of it, the host-in / host-out entry of the sequential optimiser.
"""
import numpy as np
import pytest
import torch

from helpers import load, workload_of, factors_of, rel_err, record
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine():
    import gp_mpc_amd
    eng = gp_mpc_amd.HipEngine(0)
    yield eng
    eng.close()


def _model(engine, w):
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)


def _same(a, b):
    return all(torch.equal(a[k], b[k]) for k in ("mu", "Sig", "J", "cost_mu", "cost_var"))


@pytest.mark.parametrize("N,D,A,H,tm", [(200, 3, 1, 25, False), (500, 2, 1, 12, False), (150, 3, 2, 6, True), (90, 1, 1, 5, False),
                                        (260, 2, 3, 5, False), (350, 4, 2, 5, True), (560, 3, 1, 4, False)])
def test_bitwise_equal_to_the_one_workgroup_kernel(engine, N, D, A, H, tm):
    w = synth.make_workload(N, D, A, H, 12, include_time=tm, seed=N + H)
    _model(engine, w)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    # a member keeps the per-point records of the pairs it owns items of in LDS: at D = 4 or N > 500 only cluster sizes that divide
    # the pairs well leave few enough per member (the others take the plain form, which is what the dispatch then reports)
    big = D == 4 or N > 500
    checked = 0
    for rpc in (16, 32):
        engine.set_option("rows_per_chunk", rpc)
        for B in (1, 3, 9, 12):
            engine.set_option("cluster", 1)
            engine.set_option("threads", 0)
            ref = engine.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
            assert engine.last_cluster == 1 and engine.last_rollout_path == 0
            for cs, nt in ((2, 0), (5, 1024), (8, 512), (12, 0), (16, 0), (24, 512), (32, 256)):
                if cs * ((B + 7) // 8) * 8 > 256:
                    continue
                engine.set_option("cluster", cs)
                engine.set_option("threads", nt)
                out = engine.rollout(acts[:B], w.mu0, w.S0, w.include_time, w.time0)
                assert engine.last_cluster in ((cs, 1) if big else (cs,)), (rpc, B, cs)
                assert _same(out, ref), (rpc, B, cs, nt, float((out["Sig"] - ref["Sig"]).abs().max()))
                checked += engine.last_cluster > 1
    assert checked >= (4 if big else 48), checked


@pytest.mark.parametrize("name,cs", [("traj_c2", 4), ("traj_c3", 4), ("traj_clip", 4), ("traj_constraints", 4), ("traj_bigvar", 4),
                                     ("traj_c4_time", 8)])
def test_reference_goldens_through_the_cooperative_path(engine, name, cs):
    """The cooperative form at ITS chunk length (16 / 32 rows) against the reference's trajectories: the tolerances of the plain
    path (tests/test_gpu_parity.py), including `|HIP - exact| <= |reference - exact|` where the tolerance is the north-star bound."""
    from test_gpu_parity import _check_traj, _set_cost
    g = load(name)
    w = workload_of(g)
    engine.prepare(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)
    _set_cost(engine, w, g)
    engine.set_option("cluster", cs)          # (D = 4, N = 300: a member's LDS holds the records of the few pairs it owns items of)
    out = engine.rollout(w.actions, w.mu0, w.S0, w.include_time, w.time0)
    assert engine.last_cluster == cs
    _check_traj(out, g, name, "cooperative_path")


def test_members_spread_over_the_xcds_take_the_placement_independent_exchange(engine):
    """The exchange must not depend on where the dispatcher puts the members: option `cluster_debug` 8 lays the members of a candidate
    on consecutive blocks (block b runs on XCD b % 8: the members then sit on different XCDs and the kernel's own check of
    HW_REG_XCC_ID selects the write-through form, across L2s that are not coherent with each other), 4 forces that form on one XCD.
    Same bits as the one-workgroup kernel in every case, over many steps and launches."""
    w = synth.make_workload(200, 3, 1, 25, 9, seed=11)
    _model(engine, w)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    engine.set_option("rows_per_chunk", 16)
    engine.set_option("cluster", 1)
    ref = {B: engine.rollout(acts[:B], w.mu0, w.S0) for B in (1, 9)}
    try:
        for dbg in (8, 4, 12, 0):
            engine.set_option("cluster_debug", dbg)
            for cs in (3, 8, 16):
                engine.set_option("cluster", cs)
                for rep in range(6):
                    for B in (1, 9):
                        out = engine.rollout(acts[:B], w.mu0, w.S0)
                        assert engine.last_cluster == cs
                        assert _same(out, ref[B]), (dbg, cs, rep, B)
    finally:
        engine.set_option("cluster_debug", 0)


def test_dispatch_rule(engine):
    """Few candidates of a memory with enough pairwise items take the cooperative form; small memories and large batches do not."""
    for N, H, B, expect in ((200, 25, 1, True), (200, 25, 16, True), (200, 25, 300, False), (50, 15, 1, False), (500, 10, 2, True),
                            (120, 15, 1, True), (120, 15, 40, False)):
        w = synth.make_workload(N, 3 if N != 500 else 2, 1, H, B, seed=1)
        _model(engine, w)
        engine.rollout(w.actions, w.mu0, w.S0)
        assert (engine.last_cluster > 1) == expect, (N, B, engine.last_cluster)
    engine.set_option("cluster", 1)
    w = synth.make_workload(200, 3, 1, 25, 1, seed=1)
    _model(engine, w)
    engine.rollout(w.actions, w.mu0, w.S0)
    assert engine.last_cluster == 1
    with pytest.raises(Exception):
        engine.set_option("cluster", 99)


def test_few_candidate_results_do_not_depend_on_batch_or_cluster_size(engine):
    """Lockstep restarts and the sequential optimiser both live in the few-candidate regime: one candidate alone, inside a batch
    of 5, of 16 and of 40 (cluster sizes 32 / 32 / 16 / 6 here) must give the same bits -- the chunk length depends on N only."""
    w = synth.make_workload(200, 3, 1, 25, 40, seed=4)
    _model(engine, w)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    alone = engine.rollout(acts[7:8], w.mu0, w.S0)
    sizes = {engine.last_cluster}
    for lo, hi in ((5, 10), (0, 16), (0, 40)):
        out = engine.rollout(acts[lo:hi], w.mu0, w.S0)
        sizes.add(engine.last_cluster)
        assert engine.last_cluster > 1
        assert torch.equal(out["Sig"][7 - lo], alone["Sig"][0]) and torch.equal(out["J"][7 - lo], alone["J"][0])
    assert len(sizes) >= 2
    # against the plain kernel (its own chunk length): the method's noise floor, not bits
    engine.set_option("cluster", 1)
    plain = engine.rollout(acts[7:8], w.mu0, w.S0)
    e = rel_err(alone["Sig"].cpu().numpy(), plain["Sig"].cpu().numpy())
    record("cooperative_vs_plain[c2]", Sig=e)
    assert 0.0 < e < 1e-6


def test_gradient_on_the_cooperative_forward_and_the_host_entry(engine):
    from oracle import adjoint
    g = load("lcb_grad_norm")
    w = workload_of(g)
    f = factors_of(w)
    engine.set_factors(w.X, f.iK, f.beta, w.lengthscales, w.outputscales)
    engine.set_cost(w.target, w.W, w.W_T, w.kappa)
    engine.set_option("cluster", 4)
    out = engine.rollout_grad(g["actions_model"][:1], w.mu0, w.S0, w.include_time, w.time0, trajectories=True)
    assert engine.last_cluster == 4
    assert rel_err(out["J"].cpu().numpy(), g["J"][:1]) < 1e-9
    assert rel_err(out["grad"].cpu().numpy().reshape(-1), g["grad"][0]) < 1e-7        # the reference's autograd (gp_mpc_controller.py:277)
    host = engine.objective_grad_host(g["actions_model"][0], w.mu0, w.S0, w.include_time, w.time0)
    for k in ("J", "grad", "mu", "Sig", "cost_mu", "cost_var"):
        assert np.array_equal(host[k], out[k].cpu().numpy()), k
    # a config-2-sized memory, B = 1, default dispatch: against the numpy adjoint
    w = synth.make_workload(200, 3, 1, 25, 1, seed=3)
    _model(engine, w)
    engine.set_option("cluster", 0)
    host = engine.objective_grad_host(w.actions[0], w.mu0, w.S0)
    assert engine.last_cluster > 1
    J0, g0, *_ = adjoint.lcb_and_gradient(factors_of(w), w.actions[0], w.mu0, w.S0, w.target, w.W, w.W_T, w.kappa)
    assert abs(float(host["J"][0]) - J0) < 1e-8 * abs(J0) and rel_err(host["grad"].reshape(-1), g0.reshape(-1)) < 1e-7


@pytest.mark.parametrize("N,D,A,H,tm,path", [(200, 3, 1, 25, False, "cooperative"), (60, 2, 2, 12, True, "one workgroup"),
                                               (150, 3, 2, 40, False, "sequence beyond the argument block"),
                                               (3000, 3, 1, 4, False, "streaming forward"), (90, 6, 2, 8, False, "six states"),
                                               (70, 10, 2, 5, False, "wide-state sweep")])
def test_host_entry_equals_the_device_entry(engine, N, D, A, H, tm, path):
    """gpmpc_objective_grad_host -- the sequence in the forward kernel's argument block (or its own upload launch where another
    kernel reads it first / it is too long), results copied to the pinned mirror by the reverse sweep, completion polled by the
    host -- returns bit for bit what gpmpc_rollout_grad leaves on the device, call after call with changing sequences."""
    w = synth.make_workload(N, D, A, H, 6, include_time=tm, seed=N + D)
    _model(engine, w)
    engine.set_option("cluster", 0)
    for i in (0, 3, 1, 5, 5, 2):
        host = engine.objective_grad_host(w.actions[i], w.mu0, w.S0, w.include_time, w.time0)
        out = engine.rollout_grad(w.actions[i:i + 1], w.mu0, w.S0, w.include_time, w.time0, trajectories=True)
        for k in ("J", "grad", "mu", "Sig", "cost_mu", "cost_var"):
            assert np.array_equal(host[k], out[k].cpu().numpy()), (path, i, k)
        assert np.isfinite(host["grad"]).all() and np.abs(host["grad"]).max() > 0
    if path == "cooperative":
        assert engine.last_cluster > 1
    if path == "streaming forward":
        assert engine.last_rollout_path == 1
    # a shorter horizon afterwards (the mirror and the staging arrays are per shape)
    host = engine.objective_grad_host(w.actions[0][:max(H // 2, 1)], w.mu0, w.S0, w.include_time, w.time0)
    out = engine.rollout_grad(w.actions[0:1, :max(H // 2, 1)], w.mu0, w.S0, w.include_time, w.time0, trajectories=True)
    assert np.array_equal(host["grad"], out["grad"].cpu().numpy()) and np.array_equal(host["J"], out["J"].cpu().numpy())


def test_many_launches_back_to_back(engine):
    """Exchange buffers and tags over many launches of changing shape (tags never repeat while the buffer lives)."""
    ws = [synth.make_workload(200, 3, 1, 9, 3, seed=2), synth.make_workload(300, 2, 2, 7, 9, seed=5)]
    refs = []
    for w in ws:
        _model(engine, w)
        engine.set_option("cluster", 1)
        engine.set_option("rows_per_chunk", 16)
        refs.append(engine.rollout(w.actions, w.mu0, w.S0))
    engine.set_option("cluster", 0)
    for i in range(60):
        w, ref = ws[i % 2], refs[i % 2]
        if i % 2 == 0 or i < 4:
            _model(engine, w)
        else:
            _model(engine, w)
        B = 1 + i % w.actions.shape[0]
        out = engine.rollout(w.actions[:B], w.mu0, w.S0)
        assert engine.last_cluster > 1
        assert torch.equal(out["Sig"], ref["Sig"][:B]) and torch.equal(out["J"], ref["J"][:B]), i


@pytest.mark.parametrize("spread,B", [(0, 9), (8, 1), (8, 9), (0, 1)])
def test_first_launch_of_a_fresh_handle(spread, B):
    """Regression (round 6): the tag of the placement prologue of a fresh handle's FIRST cooperative launch was 0 -- what the zeroed
    exchange buffer holds -- so members could accept the zeros as the others' XCD ids before those were published; members on
    XCD 0 agreed by luck, any other placement (candidates 1..7 of a first batch, members spread over XCDs) could disagree on the
    protocol and run into their bounded waits (NaN trajectories after ~0.4 s).  Every case starts from a handle of its own."""
    import gp_mpc_amd
    w = synth.make_workload(200, 3, 1, 12, 9, seed=21)
    acts = torch.as_tensor(w.actions, device="cuda:0")
    for cs in (3, 8):
        eng = gp_mpc_amd.HipEngine(0)
        try:
            _model(eng, w)
            eng.set_option("rows_per_chunk", 16)
            eng.set_option("cluster_debug", spread)
            eng.set_option("cluster", cs)
            out = eng.rollout(acts[:B], w.mu0, w.S0)                  # the handle's first cooperative launch
            assert eng.last_cluster == cs and bool(torch.isfinite(out["Sig"]).all())
            eng.set_option("cluster", 1)
            ref = eng.rollout(acts[:B], w.mu0, w.S0)
            assert _same(out, ref), (spread, B, cs)
        finally:
            eng.close()
