"""Candidate sharding across the GPUs of one node (one process per GPU).

Candidate action sequences are independent given the shared model and initial state, so rank r
evaluates the contiguous slice [lo_r, hi_r) with no data-path collective.  The only exchange is
the final argmin: one all_gather of (J, global index) = 16 bytes per rank over RCCL/xGMI
(latency-bound), after which every rank knows the winner and its owner; the owner broadcasts
the winning (H, A) sequence.  The reference has no distributed code; the rule reproduced here
is its sequential keep-the-best loop (rl_gp_mpc/control_objects/controllers/
gp_mpc_controller.py:146-148): lowest global index wins ties, a NaN in global slot 0 is
adopted and never displaced, any other NaN is never selected.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

LOCAL = "local"        # pass as `group` to keep a call on this process even though a process group is initialised
CEM_MAX_RECORDS = 4096      # csrc/search.hip kCemMaxB: the merge kernel sorts lists x n_elite records in LDS
_CEM_FAILED = 2147483646.0  # global-index field of the record a rank contributes when its cem_local call failed


def _active(group):
    return group is not LOCAL and dist.is_available() and dist.is_initialized()


def broadcast_from_first(array, device, group=None):
    """Rank 0's copy of a small float64 numpy array on every rank of `group` (one collective of a few doubles: RCCL for a
    GPU device, gloo on the CPU).  What the sharded searches draw from numpy's global generator -- the Philox key, the
    L-BFGS starting points -- goes through here once per control step, so the union of the slices is the single-GPU
    population whatever the launcher did about seeds."""
    a = np.ascontiguousarray(array, dtype=np.float64)
    if not _active(group):
        return a.copy()
    t = torch.as_tensor(a).to(device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy().reshape(a.shape)


def any_rank(flag, device, group=None):
    """True on every rank if `flag` is true on ANY rank of `group` (one all_reduce(MAX) of one double): how a condition only one
    rank can see -- its slice failed, its state differs from what rank 0 broadcast -- becomes a decision all ranks take together,
    so that nobody raises alone and leaves the others waiting in the next collective."""
    if not _active(group):
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item() != 0.0)


def shard_bounds(num_candidates, world_size, rank):
    """Contiguous balanced slice [lo, hi) of range(num_candidates) owned by `rank`."""
    base, extra = divmod(int(num_candidates), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(index, num_candidates, world_size):
    for r in range(world_size):
        lo, hi = shard_bounds(num_candidates, world_size, r)
        if lo <= index < hi:
            return r
    raise ValueError(index)


def combine_best(pairs):
    """pairs: per-rank (J, global_index) in rank order; index -1 = nothing selectable.
    Returns (J, index) of the global winner, or (inf, -1)."""
    best_J, best_i = math.inf, -1
    for J, i in pairs:
        i = int(i)
        if i < 0:
            continue
        if i == 0 and J != J:                  # NaN adopted in global slot 0: stays the winner
            return J, 0
        if best_i < 0 or J < best_J or (J == best_J and i < best_i):
            best_J, best_i = J, i
    return best_J, best_i


def gather_best(local_J, local_index, device, group=None):
    """all_gather of the per-rank (J, global index); returns combine_best over ranks."""
    world = dist.get_world_size(group)
    mine = torch.tensor([float(local_J), float(local_index)], dtype=torch.float64, device=device)
    flat = torch.empty(world * 2, dtype=torch.float64, device=device)     # flat: accepted by gloo and RCCL
    dist.all_gather_into_tensor(flat, mine, group=group)
    allp = flat.view(world, 2).cpu().tolist()
    return combine_best([(p[0], int(p[1])) for p in allp])


def sharded_argmin(evaluate_slice, actions_local, lo, num_candidates, device, group=None):
    """Evaluate this rank's slice and agree on the global winner.

    evaluate_slice(actions_local) -> (best_J, best_GLOBAL_index or -1) for the local slice whose
    first global index is `lo` (HipEngine.rollout + HipEngine.argmin(first_global_index=lo)).
    Returns (best_J, best_index, best_actions (H, A) tensor on `device`).
    """
    if not (dist.is_available() and dist.is_initialized()):
        J, i = evaluate_slice(actions_local)
        if i < 0:
            raise FloatingPointError("no selectable candidate (all objectives NaN)")
        return J, i, actions_local[i - lo].clone()
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    J, i = evaluate_slice(actions_local)
    J, i = gather_best(J, i, device, group)
    if i < 0:
        raise FloatingPointError("no selectable candidate (all objectives NaN)")
    owner = owner_of(i, num_candidates, world)
    H, A = actions_local.shape[1:]
    win = torch.empty((H, A), dtype=torch.float64, device=device)
    if rank == owner:
        win.copy_(actions_local[i - lo])
    src = dist.get_global_rank(group, owner) if group is not None else owner
    dist.broadcast(win, src=src, group=group)
    return J, i, win


def _local_record(engine, J, actions_local, lo, out=None):
    """Device record [best J, global index (-1: nothing selectable), winning (H*A) sequence] of this rank's slice.
    An EMPTY slice (more ranks than candidates) launches nothing and contributes (inf, -1)."""
    n, H, A = actions_local.shape
    if n == 0:
        rec = torch.zeros(2 + H * A, dtype=torch.float64, device=actions_local.device) if out is None else out.zero_()
        rec[0], rec[1] = math.inf, -1.0
        return rec
    return engine.argmin_async(J, first_global_index=lo, actions=actions_local, out=out)     # one kernel: rule + gather


def _gather_records(rec, group=None):
    """(world, flat (world * len(rec)) tensor): ONE all_gather over RCCL (gloo in the CPU tests) when a process
    group is initialised, otherwise the record itself.  Wrapped in a ROCTx range ("gpmpc_gather") on a GPU."""
    if _active(group):
        world = dist.get_world_size(group)
        flat = torch.empty(world * rec.numel(), dtype=torch.float64, device=rec.device)
        marked = rec.device.type == "cuda"
        if marked:
            torch.cuda.nvtx.range_push("gpmpc_gather")        # roctx on ROCm builds of torch
        try:
            dist.all_gather_into_tensor(flat, rec.contiguous(), group=group)
        finally:
            if marked:
                torch.cuda.nvtx.range_pop()
        return world, flat
    return 1, rec


def _winner_of(host, world, H, A):
    """Cross-rank keep-the-best rule on the gathered host records (world, 2 + H*A [+ extra])."""
    bJ, bi = combine_best([(float(host[r, 0]), int(host[r, 1])) for r in range(world)])
    if bi < 0:
        raise FloatingPointError("no selectable candidate (all objectives NaN)")
    owner = [r for r in range(world) if int(host[r, 1]) == bi][0]
    return bJ, bi, host[owner, 2:2 + H * A].view(H, A).clone()


_HOST_GROUPS = {}


def host_group(group=None):
    """A process group whose collectives run on HOST tensors (gloo) over the same ranks as `group` (None: the default
    group).  With a gloo default group that is the group itself; with RCCL it is created once (every rank must reach the
    first call together, like any new_group) and cached per ranks; an entry holds a reference to the default-group OBJECT it
    was created under and is valid only while that very object is still the default group (a strong reference: its id() cannot
    be recycled while the entry lives), so a destroyed and re-initialised default group never gets a stale gloo group."""
    if dist.get_backend(group) == "gloo":
        return dist.group.WORLD if group is None else group
    ranks = tuple(range(dist.get_world_size())) if group is None else tuple(dist.get_process_group_ranks(group))
    key = ranks
    ent = _HOST_GROUPS.get(key)
    if ent is not None and ent[1] is not dist.group.WORLD:
        ent = None
    if ent is None:
        for stale in [k for k, v in _HOST_GROUPS.items() if v[1] is not dist.group.WORLD]:
            del _HOST_GROUPS[stale]
        ent = (dist.new_group(ranks=list(ranks), backend="gloo"), dist.group.WORLD)
        _HOST_GROUPS[key] = ent
    return ent[0]


_SIDE_STREAMS = {}


def side_stream(device):
    """The stream the winner exchange runs on when it is kept off the compute stream (one per device, created on first use)."""
    idx = torch.device(device).index
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[idx]


class PendingBest:
    """Winner selection in flight: the packed per-rank records are on their way to a pinned host buffer; `result()`
    waits for THAT copy only (an event), so the host can enqueue the next batch's launches before it looks at this
    batch's winner.  With `exchange_group` set the buffer holds THIS rank's record only and `result()` exchanges the
    records between the hosts (a gloo all_gather of 16 + 8 H A bytes per rank) -- nothing of the exchange is on the GPU's
    streams.

    `result()` with an `exchange_group` is a COLLECTIVE: every rank must call it, for the same pending selections in the
    same (launch) order; the first call caches its answer, so calling it again is free and safe, skipping it on one rank
    deadlocks the others.  `host_seconds` afterwards = the host time the call spent (wait for the copy + exchange)."""

    def __init__(self, host, event, world, H, A, record=None, exchange_group=None):
        self.host, self.event, self.world, self.H, self.A = host, event, world, H, A
        self.record = record            # the device record buffer, reusable once this selection has been read
        self.exchange_group = exchange_group
        self._result = None
        self._flat = None
        self.host_seconds = None

    def result(self):
        if self._result is not None:
            return self._result
        import time
        t0 = time.perf_counter()
        if self.event is not None:
            self.event.synchronize()
        host = self.host
        if self.exchange_group is not None:
            mine = host.clone()          # gloo wants ordinary (not pinned-view) tensors on some builds
            flat = torch.empty(self.world * mine.numel(), dtype=torch.float64)
            dist.all_gather_into_tensor(flat, mine, group=self.exchange_group)
            host = flat
        self._result = _winner_of(host.view(self.world, -1), self.world, self.H, self.A)
        self._flat = None                # the gathered device tensor (side-stream exchange) is free to be reused now
        self.host_seconds = time.perf_counter() - t0
        return self._result


def select_best_async(engine, J, actions_local, lo, num_candidates, group=None, host_buffer=None, record=None,
                      exchange="rccl"):
    """select_best_on_device without the host synchronisation: local keep-the-best kernel, the packed record on its way to
    pinned host memory behind an event.  `exchange` (N > 1):
      "rccl_side" = ONE RCCL all_gather of the packed records over xGMI on a SIDE stream that waits for the record's event --
                    the compute stream never waits for the collective (BASELINE north_star: "RCCL over xGMI only for the
                    final argmin gather"); the caller must not reuse `record` before this selection has been read;
      "rccl"      = the same all_gather on the compute stream (the next launch queues behind it: ~45 us per step);
      "host"      = the copy carries this rank's record only and the records are exchanged between the hosts (gloo) when the
                    result is read (PendingBest.result) -- the fallback when no xGMI collective is wanted.
    `host_buffer` / `record`: reusable pinned / device buffers of a previous call with the same shapes.
    Returns a PendingBest."""
    n, H, A = actions_local.shape
    rec = _local_record(engine, J, actions_local, lo, out=record)
    multi = _active(group)
    if exchange == "host" and multi:
        world = dist.get_world_size(group)
        xg = host_group(group)
        if rec.device.type != "cuda":
            return PendingBest(rec.clone(), None, world, H, A, rec, exchange_group=xg)
        if host_buffer is None or host_buffer.numel() != rec.numel():
            host_buffer = torch.empty(rec.numel(), dtype=torch.float64, pin_memory=True)
        host_buffer.copy_(rec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(rec.device))
        return PendingBest(host_buffer, ev, world, H, A, rec, exchange_group=xg)
    if exchange == "rccl_side" and multi and rec.device.type == "cuda":
        world = dist.get_world_size(group)
        side = side_stream(rec.device)
        if host_buffer is None or host_buffer.numel() != world * rec.numel():
            host_buffer = torch.empty(world * rec.numel(), dtype=torch.float64, pin_memory=True)
        flat = torch.empty(world * rec.numel(), dtype=torch.float64, device=rec.device)
        # async_op: ProcessGroupNCCL runs the collective on ITS stream behind what the compute stream holds now and returns
        # without making the compute stream wait; `work.wait()` under the side stream orders only the side stream (and the
        # device-to-host copy on it) behind the collective
        work = dist.all_gather_into_tensor(flat, rec.contiguous(), group=group, async_op=True)
        with torch.cuda.stream(side):
            work.wait()
            host_buffer.copy_(flat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        # `flat` was allocated on the compute stream and is read by the side stream's copy: tell the caching allocator, or a
        # PendingBest dropped unread (an exception path) would let the block be reused while the copy is in flight
        flat.record_stream(side)
        pend = PendingBest(host_buffer, ev, world, H, A, rec)
        pend._flat = flat                                       # keep the gathered device tensor alive until read
        return pend
    world, flat = _gather_records(rec, group)
    if rec.device.type != "cuda":
        return PendingBest(flat.clone(), None, world, H, A, rec)
    if host_buffer is None or host_buffer.numel() != flat.numel():
        host_buffer = torch.empty(flat.numel(), dtype=torch.float64, pin_memory=True)
    host_buffer.copy_(flat, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(rec.device))
    return PendingBest(host_buffer, ev, world, H, A, rec)


def select_best_on_device(engine, J, actions_local, lo, num_candidates, group=None, extra=None, raise_if_none=True):
    """Device-resident variant of sharded_argmin for the HIP engine: local keep-the-best on the GPU
    (gpmpc_argmin_async), the record [J, global index, winning (H*A) sequence] packed on the device, ONE
    RCCL all_gather of 16 + 8*H*A bytes per rank, one device-to-host copy, the cross-rank rule applied on
    the host.  Returns (best_J, best_index, best_actions (H, A) host tensor).

    `extra`: an optional 1-D device tensor every rank appends to its record (same length on every rank); the
    gathered (world, len) host tensor is returned as a fourth value.  The controller ships the trajectory of the
    LAST global candidate this way (what the reference leaves in its logging caches, gp_mpc_controller.py:279-283),
    so the whole exchange of a control step stays one collective."""
    n, H, A = actions_local.shape
    rec = _local_record(engine, J, actions_local, lo)
    if extra is not None:
        rec = torch.cat([rec, extra.to(rec.device, torch.float64).reshape(-1)])
    world, flat = _gather_records(rec, group)
    host = flat.cpu().view(world, rec.numel())
    try:
        bJ, bi, win = _winner_of(host, world, H, A)
    except FloatingPointError:
        if raise_if_none:
            raise
        bJ, bi, win = math.inf, -1, None           # the caller inspects `extra` (error flags) before it decides what to raise
    if extra is None:
        return bJ, bi, win
    return bJ, bi, win, host[:, 2 + H * A:].clone()


def sharded_cem_search(engine, mu0, S0, B_total, H, A, iterations, n_elite, seed, group=None, include_time=False, time0=0.0,
                       first_candidate=None, max_change=None, action_prev=None, noise=None):
    """The device-resident cross-entropy search (HipEngine.cem_search) with its candidates sharded over the ranks: per
    iteration every rank draws and evaluates its contiguous slice of the B_total candidates (the draws are keyed by the
    GLOBAL candidate index, so the union of the slices is the single-GPU population), ONE all_gather of the slices' elite
    records (n_elite x (2 + H A) doubles per rank, RCCL on the compute stream: the refit needs them), and every rank refits
    on the union -- all ranks end with the same state, the one the single-GPU search reaches on the same draws.
    Nothing is read back between iterations.  Returns (best vector (H*A,) numpy, best J)."""
    multi = _active(group)
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    n = H * A
    if world > 1 and world * n_elite > CEM_MAX_RECORDS:
        # the merge kernel sorts world x n_elite records in LDS (4096 at most): beyond that every rank runs the whole
        # population on its own GPU -- same draws (keyed by the global candidate), same state, no collective.  The decision
        # depends on (world, n_elite) only, so all ranks take it together.
        return engine.cem_search(mu0, S0, B_total, H, A, iterations, n_elite, seed=seed, include_time=include_time, time0=time0,
                                 first_candidate=first_candidate, max_change=max_change, action_prev=action_prev, noise=noise)
    lo, hi = shard_bounds(B_total, world, rank)
    state = torch.zeros(3 * n + 1, dtype=torch.float64, device=engine.device)
    gathered = torch.zeros((world * n_elite, n + 2), dtype=torch.float64, device=engine.device) if world > 1 else None
    elites = None
    failure = None
    for it in range(int(iterations)):
        if failure is None:
            try:
                elites = engine.cem_local(mu0, S0, B_total, lo, hi - lo, H, A, it, n_elite, state, seed=seed,
                                          include_time=include_time, time0=time0, first_candidate=first_candidate,
                                          max_change=max_change, action_prev=action_prev, noise=noise, out=elites)
            except Exception as e:           # noqa: BLE001 -- handed on after the collectives, see below
                if world == 1:
                    raise
                # This rank still has to enter every all_gather of the search (the others would wait in it until the watchdog
                # fires): from here on it contributes records that sort behind every real candidate and carry a marker in
                # their index field, which all ranks see in the gathered buffer when they read the result.
                failure = e
                elites = torch.zeros((n_elite, n + 2), dtype=torch.float64, device=engine.device)
                elites[:, 0] = math.inf
                elites[:, 1] = _CEM_FAILED
        if world > 1:
            dist.all_gather_into_tensor(gathered.view(-1), elites.view(-1), group=group)
            if failure is None:
                try:
                    engine.cem_merge(gathered, n_elite, n, it, state)
                except Exception as e:       # noqa: BLE001
                    failure = e
                    elites = torch.zeros((n_elite, n + 2), dtype=torch.float64, device=engine.device)
                    elites[:, 0] = math.inf
                    elites[:, 1] = _CEM_FAILED
        else:
            engine.cem_merge(elites, n_elite, n, it, state)
    if world > 1:
        # the one synchronisation: the state and, with it, whether any rank's records of the LAST gather carry the marker (a
        # rank that failed keeps contributing marked records until the end)
        flag = (gathered[:, 1] == _CEM_FAILED).any().to(torch.float64).view(1)
        host = torch.cat([state, flag]).cpu().numpy()
        # a failure of the LAST iteration's merge is followed by no further gather that could carry the marker: one
        # all_reduce of the local failure flags, so that every rank raises (or none does)
        failed_somewhere = any_rank(failure is not None, engine.device, group)
        if failure is not None:
            raise failure
        if host[-1] != 0.0 or failed_somewhere:
            raise RuntimeError("sharded cross-entropy search: another rank's slice failed; no winner this step")
        host = host[:-1]
    else:
        host = state.cpu().numpy()                                  # the one synchronisation
    return host[2 * n:3 * n].copy(), float(host[3 * n])
