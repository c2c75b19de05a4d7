"""GpStateTransitionModel -- drop-in for rl_gp_mpc/control_objects/models/gp_model.py:39-315 whose
arithmetic runs in the HIP library (C ABI of include/gpmpc.h):

    prepare_inference   -> gpmpc_prepare   (K build, Cholesky, iK, beta; reference :182-191, 400-431)
    predict_trajectory  -> gpmpc_rollout   (H-step moment matching;     reference :60-180)

plus the batched entry points the reference lacks (`predict_trajectory_batch`,
`objective_and_gradient_batch`): B candidate action sequences per launch.

No gpytorch: the three hyper-parameters the hot path reads (lengthscale (1,E), outputscale (),
likelihood noise (1,), reference :189-190,427) live in small holder objects that keep the
reference's attribute paths and its `initialize(**{...})` contract
(controllers/gp_mpc_controller.py:223-224).  `train` (exact marginal likelihood, LBFGS, random
re-initialisation inside the constraint box; reference :193-306) keeps the reference's optimiser loop in the
spawned process the controller manages; its loss and gradient come from gpmpc_mll on the GPU.
"""
import time
from types import SimpleNamespace

import numpy as np
import torch

from .abstract_model import AbstractStateTransitionModel

F64 = torch.float64


def _t(v):
    return torch.as_tensor(np.asarray(v), dtype=F64) if not isinstance(v, torch.Tensor) else v.to(F64)


class SavedState:
    """In-memory snapshot shipped to the training process (reference gp_model.py:13-36)."""

    def __init__(self, inputs, states_change, parameters, constraints_hyperparams):
        self.inputs = inputs
        self.states_change = states_change
        self.parameters = parameters
        self.constraints_hyperparams = constraints_hyperparams

    def to_arrays(self):
        self.inputs = np.asarray(self.inputs)
        self.states_change = np.asarray(self.states_change)
        self.parameters = [{k: np.asarray(v) for k, v in p.items()} for p in self.parameters]
        self.constraints_hyperparams = {k: (v.numpy() if isinstance(v, torch.Tensor) else v)
                                        for k, v in self.constraints_hyperparams.items()}

    def to_tensors(self):
        self.inputs = _t(self.inputs)
        self.states_change = _t(self.states_change)
        self.parameters = [{k: _t(v) for k, v in p.items()} for p in self.parameters]
        self.constraints_hyperparams = {k: (_t(v) if isinstance(v, np.ndarray) else v)
                                        for k, v in self.constraints_hyperparams.items()}


class GpHyperParameters:
    """One ExactGP's hyper-parameters under the reference's attribute paths
    (covar_module.base_kernel.lengthscale (1,E), covar_module.outputscale (), likelihood.noise (1,))."""

    KEYS = ("covar_module.base_kernel.lengthscale", "covar_module.outputscale", "likelihood.noise")

    def __init__(self, lengthscale, outputscale, noise):
        self.covar_module = SimpleNamespace(base_kernel=SimpleNamespace(lengthscale=None), outputscale=None)
        self.likelihood = SimpleNamespace(noise=None)
        self.initialize(**{self.KEYS[0]: lengthscale, self.KEYS[1]: outputscale, self.KEYS[2]: noise})

    def initialize(self, **kwargs):
        for key, val in kwargs.items():
            v = _t(val)
            if key == self.KEYS[0]:
                self.covar_module.base_kernel.lengthscale = v.reshape(1, -1)
            elif key == self.KEYS[1]:
                self.covar_module.outputscale = v.reshape(())
            elif key == self.KEYS[2]:
                self.likelihood.noise = v.reshape(1)
            else:
                raise KeyError(key)
        return self

    def state_dict(self):
        return {self.KEYS[0]: self.covar_module.base_kernel.lengthscale.clone(),
                self.KEYS[1]: self.covar_module.outputscale.clone(),
                self.KEYS[2]: self.likelihood.noise.clone()}

    def eval(self):
        return self


def create_models(gp_init_dict, num_models, num_inputs):
    """Hyper-parameter holders initialised from ModelConfig.gp_init (reference :318-384)."""
    if isinstance(gp_init_dict, list):
        return [GpHyperParameters(p[GpHyperParameters.KEYS[0]], p[GpHyperParameters.KEYS[1]], p[GpHyperParameters.KEYS[2]])
                for p in gp_init_dict]
    ls = _t(gp_init_dict["base_kernel.lengthscale"])
    return [GpHyperParameters(ls[i].expand(num_inputs) if ls[i].ndim == 0 else ls[i],
                              _t(gp_init_dict["outputscale"])[i], _t(gp_init_dict["noise_covar.noise"])[i])
            for i in range(num_models)]


class TrainingFailed(list):
    """Answer of a training process that could not train: the list of the INCOMING parameter dicts (so a consumer
    that only wants parameters can use it as is) with the reason attached."""

    def __init__(self, parameters=(), reason=""):
        super().__init__(parameters)
        self.reason = reason

    def __reduce__(self):
        return (TrainingFailed, (list(self), self.reason))


class _LockstepMll:
    """Rendezvous of the per-GP optimiser threads of `train`: a thread asks for the loss of its GP and waits; when every
    unfinished GP is waiting, the last one to arrive evaluates ALL pending GPs with one gpmpc_mll call (the kernels
    batch over GPs).  If the batched call fails (a GP whose K is not positive definite at its trial point), the pending
    GPs are evaluated one by one so that only the failing GP sees the error."""

    WAIT_LIMIT = 600.0

    def __init__(self, engine, X_dev, Y_dev, n_gp):
        import threading
        self.engine, self.X, self.Y = engine, X_dev, Y_dev
        self.cond = threading.Condition()
        self.pending, self.results = {}, {}
        self.running = n_gp
        self.launches = 0

    def _flush(self):                                  # called with the lock held
        idx = sorted(self.pending)
        try:
            self._evaluate_pending(idx)
        except BaseException as e:                     # anything outside the per-GP handling below (stacking, indexing, ...):
            for i in idx:                              # every waiting GP gets a failure -- nobody may wait forever
                self.results.setdefault(i, e if isinstance(e, Exception) else RuntimeError(repr(e)))
            if not isinstance(e, Exception):           # KeyboardInterrupt / SystemExit: the waiters are served (above) and
                raise                                  # the interrupt itself goes on in the flushing thread, not swallowed
        finally:
            self.pending.clear()
            self.cond.notify_all()

    def _evaluate_pending(self, idx):
        ls = torch.stack([self.pending[i][0].reshape(-1) for i in idx])
        osc = torch.stack([self.pending[i][1].reshape(()) for i in idx])
        nz = torch.stack([self.pending[i][2].reshape(()) for i in idx])
        try:
            out = self.engine.mll(self.X, self.Y[:, idx].contiguous(), ls, osc, nz)
            self.launches += 1
            for k, i in enumerate(idx):
                self.results[i] = (float(out["loss"][k]), out["d_lengthscale"][k], float(out["d_outputscale"][k]),
                                   float(out["d_noise"][k]))
        except Exception:
            for k, i in enumerate(idx):
                try:
                    o = self.engine.mll(self.X, self.Y[:, i:i + 1].contiguous(), ls[k:k + 1], osc[k:k + 1], nz[k:k + 1])
                    self.launches += 1
                    self.results[i] = (float(o["loss"][0]), o["d_lengthscale"][0], float(o["d_outputscale"][0]),
                                       float(o["d_noise"][0]))
                except Exception as e:
                    self.results[i] = e

    def evaluate(self, a, ls, osc, nz):
        with self.cond:
            self.pending[a] = (ls, osc, nz)
            if len(self.pending) == self.running:
                self._flush()
            waited = 0.0
            while a not in self.results:
                # a rendezvous that never completes (a sibling thread died outside `finished`) must not hang the training
                # process: after WAIT_LIMIT seconds the evaluation fails and `train` reports TrainingFailed
                if not self.cond.wait(timeout=5.0):
                    waited += 5.0
                    if waited >= self.WAIT_LIMIT:
                        # withdraw the request: a stale entry would be counted by `finished` (premature flush that includes
                        # it, one wasted evaluation and a result nobody pops)
                        self.pending.pop(a, None)
                        raise TimeoutError(f"lockstep evaluation of GP {a} not served within {self.WAIT_LIMIT:.0f} s")
            r = self.results.pop(a)
        if isinstance(r, Exception):
            raise r
        return r

    def finished(self, a):
        with self.cond:
            self.running -= 1
            if self.pending and len(self.pending) == self.running:
                self._flush()


class _DeviceNegMll(torch.autograd.Function):
    """-log p(y | X, theta) / N with its gradient from gpmpc_mll (reference: gpytorch ExactMarginalLogLikelihood +
    autograd, gp_model.py:262-275); `shared` batches the evaluations of the GPs that are waiting (see _LockstepMll)."""

    @staticmethod
    def forward(ctx, ls, osc, nz, shared, a):
        loss, gl, go, gn = shared.evaluate(a, ls.detach(), osc.detach(), nz.detach())
        ctx.shapes = (ls.shape, osc.shape, nz.shape)
        ctx.grads = (gl, go, gn)
        return torch.tensor(loss, dtype=F64)

    @staticmethod
    def backward(ctx, gout):
        gl, go, gn = ctx.grads
        sl, so, sn = ctx.shapes
        return (gout * torch.as_tensor(gl, dtype=F64).reshape(sl), gout * torch.tensor(float(go), dtype=F64).reshape(so),
                gout * torch.tensor(float(gn), dtype=F64).reshape(sn), None, None)


class GpStateTransitionModel(AbstractStateTransitionModel):
    def __init__(self, config, dim_state, dim_action, engine=None, device=None):
        super().__init__(config, dim_state, dim_action)
        if self.config.include_time_model:
            self.dim_input += 1
        self.config.extend_dimensions_params(dim_state=self.dim_state, dim_input=self.dim_input)
        self.models = create_models(self.config.gp_init, self.dim_state, self.dim_input)
        self._engine = engine
        self._device = device
        self.x_mem = None
        self.y_mem = None
        self._cost_key = None

    last_training_launches = None       # gpmpc_mll calls of the last `train` in this process (lockstep: ~ the longest GP's count)

    # -- engine ------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from ...engine import HipEngine       # raises without a GPU / without the HIP library
            self._engine = HipEngine(self._device)
        return self._engine

    @property
    def lengthscales(self):
        return torch.stack([m.covar_module.base_kernel.lengthscale[0] for m in self.models])

    @property
    def variances(self):
        return torch.stack([m.covar_module.outputscale for m in self.models])

    @property
    def noises(self):
        return torch.stack([m.likelihood.noise[0] for m in self.models])

    @property
    def iK(self):
        return self.engine.factors()[0].cpu()

    @property
    def beta(self):
        return self.engine.factors()[1].cpu()

    # -- a1/a2 -------------------------------------------------------------------------
    def prepare_inference(self, inputs, state_changes):
        self.x_mem = inputs
        self.y_mem = state_changes
        self.engine.prepare(inputs, state_changes, self.lengthscales, self.variances, self.noises)

    def set_cost(self, reward_config):
        """Load the quadratic-cost / LCB settings into the engine.  The reference reads its reward config on every
        evaluation (setpoint_distance_reward_mapper.py:36-66), so in-place edits of the config must take effect:
        the (tiny) packed settings are compared by CONTENT with what this engine last received and re-sent when
        they differ -- a few hundred bytes of host work per call, no launch."""
        smin = reward_config.state_min if reward_config.use_constraints else None
        smax = reward_config.state_max if reward_config.use_constraints else None
        args = (reward_config.target_state_action_norm.numpy(), reward_config.weight_matrix_cost.numpy(),
                reward_config.weight_matrix_cost_terminal.numpy(), float(reward_config.exploration_factor),
                bool(reward_config.clip_lower_bound_cost_to_0),
                None if smin is None else smin.numpy(), None if smax is None else smax.numpy())
        key = tuple(None if a is None else (np.asarray(a, dtype=np.float64).tobytes()) for a in args)
        if getattr(self.engine, "_cost_token", None) == key:
            self._cost_key = key
            return
        self.engine.set_cost(*args)
        self._cost_key = key
        self.engine._cost_token = key          # engines can be shared between models: remember WHAT is loaded

    # -- a3/a4 -------------------------------------------------------------------------
    def predict_trajectory_batch(self, actions, obs_mu, obs_var, len_horizon=None, current_time_idx=0,
                                 trajectories=True, stage_costs=True):
        """actions (B,H,A) -> dict of DEVICE tensors: J (B,), mu (B,H+1,D), Sig (B,H+1,D,D),
        cost_mu / cost_var (B,H+1).  Costs need set_cost() first."""
        if self._cost_key is None and stage_costs:
            raise RuntimeError("call set_cost(reward_config) before predicting costs")
        actions = torch.as_tensor(np.asarray(actions) if not isinstance(actions, torch.Tensor) else actions, dtype=F64)
        if len_horizon is not None and actions.shape[1] != len_horizon:
            raise ValueError("actions.shape[1] != len_horizon")
        return self.engine.rollout(actions, _t(obs_mu).numpy(), _t(obs_var).numpy(), self.config.include_time_model,
                                   float(current_time_idx), trajectories, stage_costs)

    def objective_and_gradient_batch(self, actions, obs_mu, obs_var, current_time_idx=0, trajectories=False):
        """actions (B,H,A) -> dict of DEVICE tensors: J (B,), grad (B,H,A) = dJ/d(actions) (analytic; what the
        reference gets from autograd, gp_mpc_controller.py:277), optionally the trajectories and costs."""
        if self._cost_key is None:
            raise RuntimeError("call set_cost(reward_config) before predicting")
        actions = torch.as_tensor(np.asarray(actions) if not isinstance(actions, torch.Tensor) else actions, dtype=F64)
        return self.engine.rollout_grad(actions, _t(obs_mu).numpy(), _t(obs_var).numpy(), self.config.include_time_model,
                                        float(current_time_idx), trajectories)

    def objective_and_gradient_host(self, actions, obs_mu, obs_var, current_time_idx=0):
        """ONE sequence (H, A) -> dict of HOST (numpy) arrays J (1,), grad (1,H,A), mu, Sig, cost_mu, cost_var: the shape of call
        scipy's L-BFGS-B makes (gp_mpc_controller.py:133-141, 229-285), served by gpmpc_objective_grad_host with one
        synchronisation and no torch tensors in between."""
        if self._cost_key is None:
            raise RuntimeError("call set_cost(reward_config) before predicting")
        return self.engine.objective_grad_host(np.asarray(actions, dtype=np.float64), _t(obs_mu).numpy(), _t(obs_var).numpy(),
                                               self.config.include_time_model, float(current_time_idx))

    def predict_trajectory(self, actions, obs_mu, obs_var, len_horizon, current_time_idx):
        """Same signature / return shapes as the reference (:60-110): ((H+1,D), (H+1,D,D)) CPU tensors."""
        out = self.predict_trajectory_batch(_t(actions)[None], obs_mu, obs_var, len_horizon, current_time_idx,
                                            trajectories=True, stage_costs=False)
        return out["mu"][0].cpu(), out["Sig"][0].cpu()

    # -- state / training ------------------------------------------------------------------
    def save_state(self):
        return SavedState(inputs=self.x_mem, states_change=self.y_mem,
                          parameters=[m.state_dict() for m in self.models],
                          constraints_hyperparams={k: v for k, v in vars(self.config).items() if k != "gp_init"})

    def load_state(self, saved_state):
        for m, p in zip(self.models, saved_state.parameters):
            m.initialize(**p)
        self.prepare_inference(_t(saved_state.inputs), _t(saved_state.states_change))

    @staticmethod
    def train(queue, saved_state, lr_train, num_iter_train, clip_grad_value, print_train=False, step_print_train=25,
              device="auto", loss_evaluator=None):
        """Exact-MLL hyper-parameter search (reference :193-306): per GP a random restart inside the constraint box
        (drawn in the reference's order: outputscale, lengthscale, noise; GP after GP), LBFGS(strong_wolfe), keep the
        best, never return something worse than the incoming parameters.  Runs in the spawned training process, fp64.
        The loss and its gradient come from gpmpc_mll (K build + Cholesky + inverse + gradient contraction on the GPU)
        through an engine of this process's own; there is no CPU expression of the loss in this package.

        The D optimisations are independent, so they run as D threads advanced in LOCKSTEP: whenever every unfinished
        GP waits for a loss, ONE gpmpc_mll call evaluates all of them (the kernels batch over GPs; the reference runs
        them one after the other, :233-290).  Each GP sees the values its own sequential run would see.
        `loss_evaluator(X, y, ls, os, nz)` replaces the device loss in tests (the oracle's torch expression, as the
        checker of the driver logic); it is called per GP.

        Whatever happens in here, exactly ONE answer is put on the queue: the list of parameter dicts, or a
        `TrainingFailed` (a list holding the INCOMING parameters, with the reason attached) when the engine cannot be
        created, the device is not one this package computes on, or the search raised -- so the controller's
        check_and_close_processes never waits on a dead child and can tell a failed training from a finished one."""
        import threading
        t0 = time.time()
        incoming, out, engine, reason = None, None, None, "interrupted"
        try:
            incoming = [{k: np.asarray(v).copy() for k, v in p.items()} for p in saved_state.parameters]
            saved_state.to_tensors()
            X, Y = saved_state.inputs, saved_state.states_change
            cons = saved_state.constraints_hyperparams
            N, E = X.shape
            n_gp = len(saved_state.parameters)
            shared = None
            if loss_evaluator is None:
                if device not in ("auto", "hip"):
                    raise ValueError(f"training device {device!r}: the loss runs on the GPU only ('auto' or 'hip')")
                from ...engine import HipEngine       # raises without a GPU / without the HIP library
                engine = HipEngine(0)
                shared = _LockstepMll(engine, torch.as_tensor(X, dtype=F64).to(engine.device).contiguous(),
                                      torch.as_tensor(Y, dtype=F64).to(engine.device).contiguous(), n_gp)
            K0, K1, K2 = GpHyperParameters.KEYS
            lo = [{"ls": _t(cons["min_lengthscale"])[a], "os": _t(cons["min_outputscale"])[a],
                   "nz": _t(cons["min_std_noise"])[a] ** 2} for a in range(n_gp)]
            hi = [{"ls": _t(cons["max_lengthscale"])[a], "os": _t(cons["max_outputscale"])[a],
                   "nz": _t(cons["max_std_noise"])[a] ** 2} for a in range(n_gp)]
            start = [{"ls": p[K0].reshape(-1), "os": p[K1].reshape(()), "nz": p[K2].reshape(())}
                     for p in saved_state.parameters]
            # the restarts, drawn up front in the order the reference's sequential loop consumes the generator (:236-252)
            raws = []
            for a in range(n_gp):
                raws.append({k: torch.logit(torch.rand(start[a][k].shape, dtype=F64).clamp(1e-6, 1 - 1e-6)).requires_grad_(True)
                             for k in ("os", "ls", "nz")})
            results = [None] * n_gp

            def search(a):
                y = Y[:, a]

                def neg_mll(ls, osc, nz):
                    if shared is not None:
                        return _DeviceNegMll.apply(ls, osc, nz, shared, a)
                    return loss_evaluator(X, y, ls, osc, nz)

                best = dict(start[a])
                try:
                    best_loss = float(neg_mll(best["ls"], best["os"], best["nz"]))
                except Exception:
                    best_loss = float("inf")
                raw = raws[a]

                def val(k):
                    return lo[a][k] + (hi[a][k] - lo[a][k]) * torch.sigmoid(raw[k])
                opt = torch.optim.LBFGS([raw["ls"], raw["os"], raw["nz"]], lr=lr_train, line_search_fn="strong_wolfe")
                try:
                    for i in range(num_iter_train):
                        def closure():
                            opt.zero_grad()
                            loss = neg_mll(val("ls"), val("os"), val("nz"))
                            loss.backward()
                            return loss
                        loss = float(opt.step(closure).detach())
                        if print_train and i % step_print_train == 0:
                            print(f"train gp {a} iter {i + 1}/{num_iter_train} loss {loss:.5f}")
                        if loss < best_loss:
                            best_loss = loss
                            best = {k: val(k).detach().clone() for k in raw}
                except Exception as e:         # keep the best found so far, like the reference (:289-290)
                    print(e)
                return {K0: best["ls"].reshape(1, E).numpy(), K1: best["os"].reshape(()).numpy(),
                        K2: best["nz"].reshape(1).numpy()}

            def worker(a):
                try:
                    results[a] = search(a)
                except BaseException as e:
                    results[a] = e
                finally:
                    if shared is not None:
                        shared.finished(a)

            if shared is not None and n_gp > 1:
                threads = [threading.Thread(target=worker, args=(a,), daemon=True) for a in range(n_gp)]
                for th in threads:
                    th.start()
                for th in threads:
                    th.join()
            else:
                for a in range(n_gp):
                    worker(a)
            for r in results:
                if isinstance(r, BaseException):
                    raise r
            out = results
            if shared is not None:
                GpStateTransitionModel.last_training_launches = shared.launches
            if print_train:
                print(f"training process: {time.time() - t0:.2f} s")
        except Exception as e:
            reason = repr(e)
            print(f"training process failed ({reason}): keeping the incoming hyper-parameters")
            out = None
        finally:
            if engine is not None:
                engine.close()
            if out is not None and incoming is not None and len(out) == len(incoming):
                queue.put(out)
            else:
                keep = incoming if incoming is not None else [dict(p) for p in getattr(saved_state, "parameters", [])]
                queue.put(TrainingFailed(keep, reason))
