"""Thin object wrapper over the C ABI: device memory and streams come from PyTorch-ROCm,
all arithmetic happens in libgpmpc_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _host(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


class _HostEvaluationPlan:
    """Per (H, A): staging arrays of gpmpc_objective_grad_host's inputs, the layout of its result buffer and a view of it."""

    def __init__(self, H, A, D):
        self.actions, self.mu0, self.S0 = np.empty((H, A)), np.empty(D), np.empty((D, D))
        self.p_actions, self.p_mu0, self.p_S0 = (C.c_void_p(a.ctypes.data) for a in (self.actions, self.mu0, self.S0))
        self.res = C.POINTER(C.c_double)()
        self.res_ref = C.byref(self.res)
        self.n = 1 + H * A + (H + 1) * (D + D * D + 2)
        o = [int(v) for v in np.cumsum([0, 1, H * A, (H + 1) * D, (H + 1) * D * D, H + 1, H + 1])]
        shapes = [(1,), (1, H, A), (1, H + 1, D), (1, H + 1, D, D), (1, H + 1), (1, H + 1)]
        self.fields = [(k, o[i], o[i + 1], shapes[i]) for i, k in enumerate(("J", "grad", "mu", "Sig", "cost_mu", "cost_var"))]
        self.addr, self.view = None, None

    def result(self):
        addr = C.cast(self.res, C.c_void_p).value
        if addr != self.addr:                  # the library's pinned buffer (moves only when it grows)
            self.addr, self.view = addr, np.ctypeslib.as_array(self.res, shape=(self.n,))
        flat = self.view.copy()                # the buffer is reused by the next evaluation
        return {k: flat[lo:hi].reshape(sh) for k, lo, hi, sh in self.fields}


class HipEngine:
    """One handle per GPU (gpmpc_create).  All tensors are fp64 on ``device``."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a ROCm GPU: the GP-MPC hot path has no CPU fallback")
        self.lib = L.lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        h = C.c_void_p()
        rc = self.lib.gpmpc_create(C.byref(h), self.device.index)
        if rc != L.GPMPC_OK:
            raise L.GpmpcError(rc, "gpmpc_create failed")
        self._h = h
        self._ogh_plans = {}
        self.N = self.D = self.E = 0
        self._cost = None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gpmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------
    def _check(self, rc):
        if rc == L.GPMPC_OK:
            return
        msg = self.lib.gpmpc_last_error(self._h).decode()
        if rc == L.GPMPC_ERR_NOT_PD:
            raise L.NotPositiveDefiniteError(rc, msg)
        raise L.GpmpcError(rc, msg)

    def _dev(self, t, shape=None):
        t = torch.as_tensor(t, dtype=torch.float64).to(self.device).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_option(self, name, value):
        self._check(self.lib.gpmpc_set_option(self._h, name.encode(), int(value)))

    # -- a1/a2 -------------------------------------------------------------------------
    def prepare(self, X, Y, lengthscales, outputscales, noises):
        X = self._dev(X)
        N, E = X.shape
        Y = self._dev(Y)
        D = Y.shape[1]
        ls = self._dev(lengthscales, (D, E))
        osc = self._dev(outputscales).reshape(D)
        nz = self._dev(noises).reshape(D)
        self._keep = (X, Y, ls, osc, nz)
        self._check(self.lib.gpmpc_prepare(self._h, X.data_ptr(), Y.data_ptr(), ls.data_ptr(), osc.data_ptr(),
                                           nz.data_ptr(), N, D, E, self._stream()))
        self.N, self.D, self.E = N, D, E

    def mll(self, X, Y, lengthscales, outputscales, noises):
        """Training loss of the D GPs and its gradient (gp_model.py:262-275): dict(loss (D,), d_lengthscale (D,E),
        d_outputscale (D,), d_noise (D,)) as numpy arrays.  Replaces the cached factors of this engine."""
        X = self._dev(X)
        N, E = X.shape
        Y = self._dev(Y)
        D = Y.shape[1]
        ls = self._dev(lengthscales, (D, E))
        osc = self._dev(outputscales).reshape(D)
        nz = self._dev(noises).reshape(D)
        out = np.empty((D, E + 3), dtype=np.float64)
        self._check(self.lib.gpmpc_mll(self._h, X.data_ptr(), Y.data_ptr(), ls.data_ptr(), osc.data_ptr(), nz.data_ptr(),
                                       N, D, E, out.ctypes.data_as(C.POINTER(C.c_double)), self._stream()))
        self.N, self.D, self.E = N, D, E
        return {"loss": out[:, 0].copy(), "d_lengthscale": out[:, 1:1 + E].copy(), "d_outputscale": out[:, 1 + E].copy(),
                "d_noise": out[:, 2 + E].copy()}

    @property
    def last_prepare_mode(self):
        """0 = full factorisation, 1 = border update of the cached factors, 2 = cache hit."""
        return int(self.lib.gpmpc_last_prepare_mode(self._h))

    @property
    def last_rollout_path(self):
        """Kernels of the last forward pass: 0 = fused-horizon kernel, 1 = streaming kernel, 2 = batch-major tiles."""
        return int(self.lib.gpmpc_last_rollout_path(self._h))

    @property
    def last_cluster(self):
        """Workgroups per candidate of the last fused-horizon launch (1 = the plain kernel; > 1 = the few-candidate cooperative form)."""
        return int(self.lib.gpmpc_last_cluster(self._h))

    @property
    def last_grad_path(self):
        """Moment passes of the last `rollout_grad` (bit mask): 1 separable off-diagonal pairs, 2 tile moments of the diagonal
        pairs, 4 streaming element-wise pass, 8 the 8 < D <= 16 pass, 16 tile moments formed inside the batch-major forward."""
        return int(self.lib.gpmpc_last_grad_path(self._h))

    @property
    def build_id(self):
        """Hash over the sources the loaded library was built from (gpmpc_build_id)."""
        return self.lib.gpmpc_build_id().decode()

    def set_factors(self, X, iK, beta, lengthscales, outputscales):
        X = self._dev(X)
        N, E = X.shape
        beta = self._dev(beta)
        D = beta.shape[0]
        iK = self._dev(iK, (D, N, N))
        ls = self._dev(lengthscales, (D, E))
        osc = self._dev(outputscales).reshape(D)
        self._check(self.lib.gpmpc_set_factors(self._h, X.data_ptr(), iK.data_ptr(), beta.data_ptr(), ls.data_ptr(),
                                               osc.data_ptr(), N, D, E, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        self.N, self.D, self.E = N, D, E

    def factors(self):
        iK = torch.empty((self.D, self.N, self.N), dtype=torch.float64, device=self.device)
        beta = torch.empty((self.D, self.N), dtype=torch.float64, device=self.device)
        self._check(self.lib.gpmpc_read_factors(self._h, iK.data_ptr(), beta.data_ptr(), self._stream()))
        return iK, beta

    # -- a6 ----------------------------------------------------------------------------
    def set_cost(self, target, W, W_T, kappa, clip_to_zero=False, state_min=None, state_max=None):
        W_T = _host(W_T)
        D = W_T.shape[0]
        W = _host(W)
        A = W.shape[0] - D
        target = _host(target, (D + A,))
        smin = _host(state_min, (D,)) if state_min is not None else None
        smax = _host(state_max, (D,)) if state_max is not None else None
        self._cost_token = None        # whoever caches "my settings are loaded" (GpStateTransitionModel.set_cost) must re-key
        self._check(self.lib.gpmpc_set_cost(self._h, _hp(target), _hp(W), _hp(W_T), float(kappa), int(bool(clip_to_zero)),
                                            _hp(smin) if smin is not None else None,
                                            _hp(smax) if smax is not None else None, D, A))
        self._cost = (D, A)

    # -- a3-a5 -------------------------------------------------------------------------
    def rollout(self, actions, mu0, S0, include_time=False, time0=0.0, trajectories=True, stage_costs=True, out=None):
        """actions (B,H,A) -> dict(J (B,), [mu (B,H+1,D), Sig (B,H+1,D,D)], [cost_mu, cost_var (B,H+1)]).
        `out` = the dict of a previous call with the same shapes: its tensors are overwritten (no allocation)."""
        actions = self._dev(actions)
        B, H, A = actions.shape
        D = self.D
        mu0 = _host(mu0, (D,))
        S0 = _host(S0, (D, D))
        if out is None:
            out = {}
            # the objective needs gpmpc_set_cost FOR THIS (D, A); a trajectory-only call passes no cost pointer at all
            if stage_costs or self._cost == (D, A):
                out["J"] = torch.empty(B, dtype=torch.float64, device=self.device)
            if trajectories:
                out["mu"] = torch.empty((B, H + 1, D), dtype=torch.float64, device=self.device)
                out["Sig"] = torch.empty((B, H + 1, D, D), dtype=torch.float64, device=self.device)
            if stage_costs:
                out["cost_mu"] = torch.empty((B, H + 1), dtype=torch.float64, device=self.device)
                out["cost_var"] = torch.empty((B, H + 1), dtype=torch.float64, device=self.device)
        elif ("J" in out and out["J"].shape != (B,)) or ("mu" in out and out["mu"].shape != (B, H + 1, D)):
            raise ValueError("`out` does not match the batch shape")

        def ptr(k):
            return out[k].data_ptr() if k in out else None
        self._check(self.lib.gpmpc_rollout(self._h, actions.data_ptr(), _hp(mu0), _hp(S0), B, H, A, int(bool(include_time)),
                                           float(time0), ptr("mu"), ptr("Sig"), ptr("cost_mu"), ptr("cost_var"),
                                           ptr("J"), self._stream()))
        return out

    def rollout_grad(self, actions, mu0, S0, include_time=False, time0=0.0, trajectories=False):
        """Objective and analytic gradient: dict(J (B,), grad (B,H,A) = dJ/d(actions), [mu, Sig, cost_mu, cost_var]).
        Raises GpmpcError(GPMPC_ERR_LIMIT) for shapes the gradient kernels do not cover (callers then
        difference `rollout`)."""
        actions = self._dev(actions)
        B, H, A = actions.shape
        D = self.D
        mu0 = _host(mu0, (D,))
        S0 = _host(S0, (D, D))
        # one allocation, views into it: a caller that wants everything on the host needs ONE copy (`host_views`)
        shapes = [("J", (B,)), ("grad", (B, H, A))]
        if trajectories:
            shapes += [("mu", (B, H + 1, D)), ("Sig", (B, H + 1, D, D)), ("cost_mu", (B, H + 1)), ("cost_var", (B, H + 1))]
        sizes = [int(np.prod(sh)) for _, sh in shapes]
        pack = torch.empty(sum(sizes), dtype=torch.float64, device=self.device)
        out, off = {"packed": pack, "layout": []}, 0
        for (k, sh), n in zip(shapes, sizes):
            out[k] = pack[off:off + n].view(sh)
            out["layout"].append((k, sh, off, n))
            off += n
        self._check(self.lib.gpmpc_rollout_grad(self._h, actions.data_ptr(), _hp(mu0), _hp(S0), B, H, A,
                                                int(bool(include_time)), float(time0), out["J"].data_ptr(),
                                                out["grad"].data_ptr(),
                                                out["mu"].data_ptr() if trajectories else None,
                                                out["Sig"].data_ptr() if trajectories else None,
                                                out["cost_mu"].data_ptr() if trajectories else None,
                                                out["cost_var"].data_ptr() if trajectories else None, self._stream()))
        return out

    def objective_grad_host(self, actions, mu0, S0, include_time=False, time0=0.0):
        """ONE action sequence (H, A) on the host -> objective, gradient, trajectory and stage costs on the host
        (gpmpc_objective_grad_host: the sequence travels as a kernel argument, the results through a pinned host buffer,
        one synchronisation) -- what a host-side optimiser that evaluates one sequence per call needs (the reference's
        scipy L-BFGS-B loop, gp_mpc_controller.py:133-141).  Returns numpy arrays (copies: the buffer is reused)."""
        actions = np.asarray(actions, dtype=np.float64)
        H, A = actions.shape
        plan = self._ogh_plans.get((H, A))
        if plan is None:
            plan = self._ogh_plans[(H, A)] = _HostEvaluationPlan(H, A, self.D)
        # (staging arrays with known addresses: taking an array's address through ctypes costs more than copying 25 numbers)
        np.copyto(plan.actions, actions)
        np.copyto(plan.mu0, np.asarray(mu0, dtype=np.float64).reshape(plan.mu0.shape))
        np.copyto(plan.S0, np.asarray(S0, dtype=np.float64).reshape(plan.S0.shape))
        self._check(self.lib.gpmpc_objective_grad_host(self._h, plan.p_actions, plan.p_mu0, plan.p_S0, H, A, int(bool(include_time)),
                                                       float(time0), plan.res_ref, self._stream()))
        return plan.result()

    @staticmethod
    def host_views(out):
        """All tensors of a `rollout_grad` result on the host with a single device-to-host copy."""
        host = out["packed"].cpu()
        return {k: host[off:off + n].view(sh) for k, sh, off, n in out["layout"]}

    def rollout_timed(self, actions, mu0, S0, reps, include_time=False, time0=0.0):
        """Average kernel milliseconds per launch, measured with HIP events on the launch stream."""
        actions = self._dev(actions)
        B, H, A = actions.shape
        mu0 = _host(mu0, (self.D,))
        S0 = _host(S0, (self.D, self.D))
        J = torch.empty(B, dtype=torch.float64, device=self.device)
        ms = C.c_float(0.0)
        self._check(self.lib.gpmpc_rollout_timed(self._h, actions.data_ptr(), _hp(mu0), _hp(S0), B, H, A,
                                                 int(bool(include_time)), float(time0), J.data_ptr(), int(reps),
                                                 C.byref(ms), self._stream()))
        return float(ms.value), J

    def cem_search(self, mu0, S0, B, H, A, iterations, n_elite, seed=0, include_time=False, time0=0.0, first_candidate=None,
                   max_change=None, action_prev=None, noise=None):
        """Cross-entropy search over [0,1]^(H*A) with the whole loop on the device (gpmpc_cem_search): returns
        (best optimiser vector (H*A,) numpy, best J).  `max_change` / `action_prev` given => DerivativeActionMapper,
        else the identity mapper.  `noise` (iterations, B, H*A) device / host tensor replaces the Philox draws (tests).
        ONE host synchronisation, at the end."""
        D = self.D
        mu0 = _host(mu0, (D,))
        S0 = _host(S0, (D, D))
        n = H * A
        best = torch.empty(n + 1, dtype=torch.float64, device=self.device)
        first = _host(first_candidate, (n,)) if first_candidate is not None else None
        mapper = 0 if max_change is None else 1
        mc = _host(max_change, (A,)) if mapper else None
        ap = _host(action_prev, (A,)) if mapper else None
        nz = None if noise is None else self._dev(noise, (iterations, B, n))
        self._check(self.lib.gpmpc_cem_search(self._h, _hp(mu0), _hp(S0), B, H, A, int(bool(include_time)), float(time0),
                                              int(iterations), int(n_elite), int(seed) & (2 ** 64 - 1),
                                              _hp(first) if first is not None else None, mapper,
                                              _hp(mc) if mapper else None, _hp(ap) if mapper else None,
                                              nz.data_ptr() if nz is not None else None, best.data_ptr(), self._stream()))
        host = best.cpu().numpy()                               # the one synchronisation
        return host[:n].copy(), float(host[n])

    def cem_local(self, mu0, S0, B_total, first, B_local, H, A, iteration, n_elite, state, seed=0, include_time=False,
                  time0=0.0, first_candidate=None, max_change=None, action_prev=None, noise=None, out=None):
        """One iteration of the cross-entropy search for the slice [first, first + B_local) of B_total candidates
        (gpmpc_cem_local): returns the slice's elite records (n_elite, 2 + H*A) on the device.  `state` = the device tensor
        [mean | std | best | best J] (3 H A + 1) `cem_merge` maintains.  No synchronisation."""
        D = self.D
        mu0 = _host(mu0, (D,))
        S0 = _host(S0, (D, D))
        n = H * A
        if out is None:
            out = torch.empty((n_elite, n + 2), dtype=torch.float64, device=self.device)
        first_c = _host(first_candidate, (n,)) if first_candidate is not None else None
        mapper = 0 if max_change is None else 1
        mc = _host(max_change, (A,)) if mapper else None
        ap = _host(action_prev, (A,)) if mapper else None
        nz = None if noise is None else self._dev(noise)
        if nz is not None and (nz.dim() != 3 or nz.shape[1] != B_total or nz.shape[2] != n):
            raise ValueError("noise must be (iterations, B_total, H*A)")
        self._check(self.lib.gpmpc_cem_local(self._h, _hp(mu0), _hp(S0), int(B_total), int(first), int(B_local), H, A,
                                             int(bool(include_time)), float(time0), int(iteration), int(n_elite),
                                             int(seed) & (2 ** 64 - 1), _hp(first_c) if first_c is not None else None, mapper,
                                             _hp(mc) if mapper else None, _hp(ap) if mapper else None,
                                             nz.data_ptr() if nz is not None else None, state.data_ptr(), out.data_ptr(),
                                             self._stream()))
        self._keep_cem = (nz, state)
        return out

    def cem_merge(self, elites, n_elite, n, iteration, state):
        """Refit on the union of the slices' elite records (lists * n_elite, 2 + n) (gpmpc_cem_merge); updates `state`."""
        lists = elites.numel() // (n_elite * (n + 2))
        self._check(self.lib.gpmpc_cem_merge(self._h, elites.data_ptr(), lists, int(n_elite), int(n), int(iteration),
                                             state.data_ptr(), self._stream()))

    # -- a8 ----------------------------------------------------------------------------
    def argmin_async(self, J, first_global_index=0, actions=None, out=None):
        """Device-side keep-the-best, no host synchronisation: returns the device record
        [best J, global index as double (-1: none), winning action sequence (H*A values, if `actions` given)]."""
        J = self._dev(J)
        ha = 0 if actions is None else int(actions[0].numel())
        if out is None:
            out = torch.empty(2 + ha, dtype=torch.float64, device=self.device)
        self._check(self.lib.gpmpc_argmin_async(self._h, J.data_ptr(), J.numel(), int(first_global_index),
                                                None if actions is None else actions.data_ptr(), ha,
                                                out.data_ptr(), self._stream()))
        return out

    def argmin(self, J, first_global_index=0):
        """Keep-the-best rule over J; returns (best_J, GLOBAL index) -- index -1 if nothing selectable."""
        J = self._dev(J)
        bj = C.c_double()
        bi = C.c_longlong()
        self._check(self.lib.gpmpc_argmin(self._h, J.data_ptr(), J.numel(), int(first_global_index), C.byref(bj),
                                          C.byref(bi), self._stream()))
        return float(bj.value), int(bi.value)
