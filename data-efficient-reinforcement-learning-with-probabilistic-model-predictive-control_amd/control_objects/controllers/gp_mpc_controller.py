"""GpMpcController -- drop-in for rl_gp_mpc/control_objects/controllers/gp_mpc_controller.py.

Same constructor and method set as the reference (what run_env_function.py:18-47 calls):
get_action, get_iter_info, compute_cost_unnormalized, add_memory, check_and_close_processes,
compute_mean_lcb_trajectory; `compute_action` is an alias of get_action.

What changes is WHERE the work happens.  The reference evaluates one action sequence per Python
call (:229-285) inside a sequential restart loop (:125-148).  Here every evaluation is a batch in
ONE launch of the HIP rollout kernel:

  * optimize=False (random shooting, :142-144): all `restarts_optim` candidates at once, argmin on
    the device, candidates optionally sharded across GPUs (sharding.py);
  * optimize=True (scipy L-BFGS-B with jac=True, :132-141): value + ANALYTIC gradient per evaluation
    (gpmpc_rollout_grad: forward rollout + pairwise moment pass + reverse sweep, what the reference gets
    from `mean_cost.backward()`, :277).  Shapes outside the gradient kernels (GPMPC_ERR_LIMIT) fall
    back to a 4th-order central difference over the H*A model actions, 4*H*A + 1 candidates in one
    launch -- in `compute_mean_lcb_trajectory` and in the batched `objective_and_gradient_batch` alike.
    The reference's two pass-through clamps (action clamp of DerivativeActionMapper, optional clip of
    the UCB at 0) have identity backward; the finite difference is taken where those clamps are
    transparent so the returned gradient has the same meaning.
"""
import multiprocessing

import numpy as np
import torch
from scipy.optimize import minimize

from ..actions_mappers.action_init_functions import (generate_mpc_action_init_frompreviousiter,
                                                     generate_mpc_action_init_random)
from ..actions_mappers.mappers import DerivativeActionMapper, NormalizationActionMapper
from ..memories.gp_memory import Memory
from ..models.gp_model import GpStateTransitionModel, TrainingFailed
from ..observations_states_mappers.normalization_observation_state_mapper import NormalizationObservationStateMapper
from ..states_reward_mappers.setpoint_distance_reward_mapper import SetpointStateRewardMapper
from .abstract_controller import BaseControllerObject
from .iteration_info_class import IterationInformation
from ..._lib import GPMPC_ERR_LIMIT, GpmpcError

F64 = torch.float64
FD_STEP = 1e-3           # 4th-order stencil: truncation ~ h^4, rounding ~ 1e-13 / h


class GpMpcController(BaseControllerObject):
    def __init__(self, observation_low, observation_high, action_low, action_high, config, engine=None, device=None):
        self.config = config
        self.observation_state_mapper = NormalizationObservationStateMapper(
            config=config.observation, observation_low=observation_low, observation_high=observation_high)
        Mapper = DerivativeActionMapper if config.actions.limit_action_change else NormalizationActionMapper
        self.actions_mapper = Mapper(config=config.actions, action_low=action_low, action_high=action_high,
                                     len_horizon=config.controller.len_horizon)
        self.transition_model = GpStateTransitionModel(config=config.model,
                                                       dim_state=self.observation_state_mapper.dim_observation,
                                                       dim_action=self.actions_mapper.dim_action,
                                                       engine=engine, device=device)
        self.state_reward_mapper = SetpointStateRewardMapper(config=config.reward)
        self.memory = Memory(config.memory, dim_input=self.transition_model.dim_input,
                             dim_state=self.transition_model.dim_state,
                             include_time_model=self.transition_model.config.include_time_model,
                             step_model=config.controller.num_repeat_actions)
        self.actions_mpc_previous_iter = None
        self.iter_ctrl = 0
        self.num_cores_main = multiprocessing.cpu_count()
        self.ctx = multiprocessing.get_context("spawn")
        self.queue_train = self.ctx.Queue()
        self.info_iters = {}
        self.num_rollouts = 0          # candidate trajectories evaluated so far (throughput accounting)
        self.analytic_gradient = True  # gradient kernels (gpmpc_rollout_grad); False: 4th-order differences of the rollout
        self.process_group = None      # torch.distributed group the candidates shard over (None: the default group)

    def _ranks(self):
        """(world, rank) of the candidate sharding.  Sharding is OPT-IN (`ControllerConfig.shard_over_ranks`): a process that
        initialised torch.distributed for some other purpose must not have its candidates silently split over ranks that may
        sit at different states.  With the flag set and a process group initialised, every rank must call get_action with the
        SAME observation and the same numpy seed -- both are verified inside the exchanges (a mismatch raises on every rank)."""
        dist = torch.distributed
        if not getattr(self.config.controller, "shard_over_ranks", False) or not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self.process_group), dist.get_rank(self.process_group)

    @staticmethod
    def _state_checksum(state_mu, state_var):
        """One double that differs between ranks whose (state, state variance) differ: travels inside the winner records."""
        a = np.concatenate([np.asarray(state_mu, dtype=np.float64).ravel(), np.asarray(state_var, dtype=np.float64).ravel()])
        w = np.cos(np.arange(1, a.size + 1, dtype=np.float64))           # position-dependent weights: permutations differ too
        return float(np.dot(a, w))

    @staticmethod
    def _assert_ranks_agree(column, what):
        if not bool((column == column[0]).all()):
            raise RuntimeError(f"candidate sharding: the ranks disagree on {what} ({column.tolist()}); every rank must see the "
                               "same observation and seed numpy identically (or switch ControllerConfig.shard_over_ranks off)")

    # ------------------------------------------------------------------------------ public
    def get_action(self, obs_mu, obs_var=None, random=False):
        """Reference :52-112.  Returns the raw (denormalised) action for the environment, shape (A,)."""
        self.check_and_close_processes()
        cc = self.config.controller
        if self.iter_ctrl % cc.num_repeat_actions == 0:
            self.memory.prepare_for_model()
            state_mu, state_var = self.observation_state_mapper.get_state(obs=obs_mu, obs_var=obs_var, update_internals=True)
            actions_model = self._get_random_actions(state_mu, state_var) if random \
                else self._get_optimal_actions(state_mu, state_var)
            actions_raw = self.actions_mapper.transform_action_model_to_action_raw(actions_model, update_internals=True)
            next_action_raw = actions_raw[0]
            reward, reward_var = self.state_reward_mapper.get_reward(state_mu, state_var, actions_model[0])
            std_pred = torch.diagonal(self.states_var_pred, dim1=-2, dim2=-1).sqrt()
            idx_pred = np.arange(self.iter_ctrl, self.iter_ctrl + cc.len_horizon * cc.num_repeat_actions,
                                 cc.num_repeat_actions)
            self.iter_info = IterationInformation(
                iteration=self.iter_ctrl, state=self.states_mu_pred[0], cost=-reward.item(),
                cost_std=reward_var.sqrt().item(),
                mean_predicted_cost=np.min([-self.rewards_trajectory.mean().item(), 3]),
                mean_predicted_cost_std=self.rewards_traj_var.sqrt().mean().item(),
                lower_bound_mean_predicted_cost=self.cost_traj_mean_lcb.item(),
                predicted_idxs=idx_pred, predicted_states=self.states_mu_pred, predicted_states_std=std_pred,
                predicted_actions=actions_model, predicted_costs=-self.rewards_trajectory,
                predicted_costs_std=self.rewards_traj_var.sqrt())
            self.store_iter_info(self.iter_info)
            self.past_action = next_action_raw
        else:
            next_action_raw = self.past_action
        self.iter_ctrl += 1
        return next_action_raw.detach().cpu().numpy().copy() if isinstance(next_action_raw, torch.Tensor) \
            else np.array(next_action_raw)

    compute_action = get_action      # the name BASELINE.json's north_star uses

    def evaluate_candidates(self, actions_mpc_batch, obs_mu, obs_var, trajectories=False):
        """Objective of B optimiser vectors (B, H*A) in one launch -> dict of device tensors + 'actions_model'."""
        acts = self.actions_mapper.mpc_to_model_batch(np.asarray(actions_mpc_batch, dtype=np.float64))
        self.transition_model.set_cost(self.config.reward)
        out = self.transition_model.predict_trajectory_batch(
            acts, obs_mu, obs_var, self.config.controller.len_horizon, self.iter_ctrl,
            trajectories=trajectories, stage_costs=True)
        self.num_rollouts += acts.shape[0]
        out["actions_model"] = acts
        return out

    def compute_mean_lcb_trajectory(self, actions_mpc, obs_mu, obs_var):
        """Reference :229-285: (mean-LCB cost, d cost / d actions_mpc) for ONE optimiser vector; also caches
        the predicted trajectory and its costs on `self` for IterationInformation (:279-283)."""
        H, A = self.config.controller.len_horizon, self.actions_mapper.dim_action
        base = self.actions_mapper.mpc_to_model_batch(np.asarray(actions_mpc, dtype=np.float64).reshape(1, -1))[0]
        n = H * A
        self.transition_model.set_cost(self.config.reward)
        if self.analytic_gradient:
            try:
                host = self.transition_model.objective_and_gradient_host(base, obs_mu, obs_var, self.iter_ctrl)
            except GpmpcError as e:
                if e.code != GPMPC_ERR_LIMIT:
                    raise
                self.analytic_gradient = False         # shape outside the gradient kernels: difference the rollout
            else:
                self.num_rollouts += 1
                grad = self.actions_mapper.chain_grad_model_to_mpc(host["grad"][0])
                self._lazy_host = host                          # the caches of :279-283, made on first read
                return float(host["J"][0]), grad
        J, g_model, out = self._objective_and_gradient_by_differences(base[None], obs_mu, obs_var, trajectories=True)
        grad = self.actions_mapper.chain_grad_model_to_mpc(g_model[0])
        self._cache_trajectory(out, 0)
        return float(J[0]), grad

    def _objective_and_gradient_by_differences(self, bases, obs_mu, obs_var, trajectories=False):
        """Fallback for shapes outside the gradient kernels (GPMPC_ERR_LIMIT): J and dJ/d(model actions) of `bases`
        (C, H, A) from a 4th-order central difference of the rollout, all C * (4*H*A + 1) candidates in ONE launch
        ([base, +h, -h, +2h, -2h] per coordinate; the base candidates come first, so out[...][c] is candidate c)."""
        C, H, A = bases.shape
        n = H * A
        cand = np.empty((C, 4 * n + 1, n))
        cand[:] = bases.reshape(C, 1, n)
        k = np.arange(n)
        cand[:, 1 + k, k] += FD_STEP
        cand[:, 1 + n + k, k] -= FD_STEP
        cand[:, 1 + 2 * n + k, k] += 2 * FD_STEP
        cand[:, 1 + 3 * n + k, k] -= 2 * FD_STEP
        order = np.concatenate([np.arange(C) * (4 * n + 1)] + [c * (4 * n + 1) + 1 + np.arange(4 * n) for c in range(C)])
        flat = cand.reshape(C * (4 * n + 1), H, A)[order]
        self.transition_model.set_cost(self.config.reward)
        out = self.transition_model.predict_trajectory_batch(flat, obs_mu, obs_var, H, self.iter_ctrl,
                                                             trajectories=trajectories, stage_costs=True)
        self.num_rollouts += flat.shape[0]
        cm = out["cost_mu"].cpu().numpy()
        cv = out["cost_var"].cpu().numpy()
        J_clip = out["J"].cpu().numpy()[:C]
        J_free = np.mean(cm - self.config.reward.exploration_factor * np.sqrt(cv), axis=1)   # clamp transparent
        Jd = J_free[C:].reshape(C, 4, n)
        g_model = (8.0 * (Jd[:, 0] - Jd[:, 1]) - (Jd[:, 2] - Jd[:, 3])) / (12.0 * FD_STEP)
        return J_clip, g_model.reshape(C, H, A), out

    def compute_cost_unnormalized(self, obs, action, obs_var=None):
        """Reference :287-305: cost of an un-normalised (obs, action) pair -> (mean, variance)."""
        state_mu, state_var = self.observation_state_mapper.get_state(obs=obs, obs_var=obs_var, update_internals=False)
        action_model = self.actions_mapper.transform_action_raw_to_action_model(action)
        r, v = self.state_reward_mapper.get_reward(state_mu, state_var, action_model)
        return -r.item(), v.item()

    def add_memory(self, obs, action, obs_new, reward, predicted_state=None, predicted_state_std=None):
        """Reference :165-199."""
        state_mu, _ = self.observation_state_mapper.get_state(obs=obs, update_internals=False)
        state_mu_new, _ = self.observation_state_mapper.get_state(obs=obs_new, update_internals=False)
        action_model = self.actions_mapper.transform_action_raw_to_action_model(action)
        self.memory.add(state_mu, action_model, state_mu_new, reward, iter_ctrl=self.iter_ctrl - 1,
                        predicted_state=predicted_state, predicted_state_std=predicted_state_std)
        training_idle = not (hasattr(self, "p_train") and not self.p_train._closed)
        if self.iter_ctrl % self.config.training.training_frequency == 0 and training_idle:
            self.start_training_process()

    def start_training_process(self):
        saved_state = self.transition_model.save_state()
        saved_state.to_arrays()
        tc = self.config.training
        self.p_train = self.ctx.Process(target=GpStateTransitionModel.train,
                                        args=(self.queue_train, saved_state, tc.lr_train, tc.iter_train,
                                              tc.clip_grad_value, tc.print_train, tc.step_print_train,
                                              getattr(tc, "device", "auto")))
        self.p_train.start()

    def check_and_close_processes(self):
        """Reference :216-227: collect finished training, load the new hyper-parameters, refactorise."""
        if hasattr(self, "p_train") and not self.p_train._closed and not self.p_train.is_alive():
            # train() always queues exactly one result; a child that died before it could (killed, import error)
            # must not block the control loop: wait briefly, then keep the current hyper-parameters
            import queue as _queue
            try:
                params = self.queue_train.get(timeout=5.0 if self.p_train.exitcode == 0 else 0.5)
            except _queue.Empty:
                print(f"training process ended with exit code {self.p_train.exitcode} and no result: "
                      "keeping the current hyper-parameters")
                params = None
            self.p_train.join()
            if isinstance(params, TrainingFailed):
                # the child answered, but could not train (no GPU for it, unsupported device, search raised): say so
                # instead of silently carrying stale hyper-parameters (they are the incoming ones)
                self.training_failures = getattr(self, "training_failures", 0) + 1
                print(f"GP training failed ({params.reason}): keeping the current hyper-parameters "
                      f"({self.training_failures} failed training(s) so far)")
                params = None
            if params is not None:
                for model, p in zip(self.transition_model.models, params):
                    model.initialize(**p)
            self.p_train.close()
            x_mem, y_mem = self.memory.get()
            self.transition_model.prepare_inference(x_mem, y_mem)

    def get_iter_info(self):
        return self.iter_info

    def store_iter_info(self, iter_info):
        for key, val in vars(iter_info).items():
            self.info_iters.setdefault(key, []).append(val)

    # ---------------------------------------------------------------------------- internals
    # The logging caches of the reference (:279-283) are written by EVERY objective evaluation but read once per control step
    # (get_action, :88-106): the sequential optimiser's evaluations keep the host arrays of their last call and the five
    # tensors are made when somebody looks.
    def _materialise_caches(self):
        host = self.__dict__.pop("_lazy_host", None)
        if host is not None:
            self._cache_trajectory({k: torch.from_numpy(v) for k, v in host.items()}, 0)

    def _lazy_cache_property(name):                    # noqa: N805 -- class-body helper
        def get(self):
            self._materialise_caches()
            return self.__dict__[name]

        def put(self, value):
            self.__dict__.pop("_lazy_host", None) if name == "states_mu_pred" else None
            self.__dict__[name] = value
        return property(get, put)

    states_mu_pred = _lazy_cache_property("states_mu_pred")
    states_var_pred = _lazy_cache_property("states_var_pred")
    rewards_trajectory = _lazy_cache_property("rewards_trajectory")
    rewards_traj_var = _lazy_cache_property("rewards_traj_var")
    cost_traj_mean_lcb = _lazy_cache_property("cost_traj_mean_lcb")
    del _lazy_cache_property

    def _cache_trajectory(self, out, idx):
        self.states_mu_pred = out["mu"][idx].cpu()
        self.states_var_pred = out["Sig"][idx].cpu()
        self.rewards_trajectory = -out["cost_mu"][idx].cpu()
        self.rewards_traj_var = out["cost_var"][idx].cpu()
        self.cost_traj_mean_lcb = -out["J"][idx].cpu()

    def _cache_packed_trajectory(self, packed, H, D):
        """The same caches from the [mu, Sig, cost_mu, cost_var, J] record another rank shipped (sharded shooting)."""
        n1, n2, n3 = (H + 1) * D, (H + 1) * D * D, H + 1
        o = np.cumsum([0, n1, n2, n3, n3])
        self.states_mu_pred = packed[o[0]:o[1]].view(H + 1, D).clone()
        self.states_var_pred = packed[o[1]:o[2]].view(H + 1, D, D).clone()
        self.rewards_trajectory = -packed[o[2]:o[3]].clone()
        self.rewards_traj_var = packed[o[3]:o[4]].clone()
        self.cost_traj_mean_lcb = -packed[o[4]].clone()

    def _prepare(self):
        x_mem, y_mem = self.memory.get()
        self.transition_model.prepare_inference(x_mem, y_mem)

    def _get_optimal_actions(self, state_mu, state_var):
        """Reference :114-153."""
        self._prepare()
        cc = self.config.controller
        H, A = cc.len_horizon, self.actions_mapper.dim_action
        if not cc.optimize:
            return self._random_shooting(state_mu, state_var)
        if getattr(cc, "candidate_optimizer", None) == "cem":
            return self._cross_entropy_search(state_mu, state_var)
        if getattr(cc, "candidate_optimizer", None) == "cem_device":
            return self._device_cross_entropy_search(state_mu, state_var)
        if getattr(cc, "candidate_optimizer", None) == "lbfgs":
            return self._batched_lbfgs_search(state_mu, state_var)
        opt_fun, best = np.inf, None
        for idx_restart in range(cc.restarts_optim):
            if cc.init_from_previous_actions and self.actions_mpc_previous_iter is not None and idx_restart == 0:
                x0 = generate_mpc_action_init_frompreviousiter(self.actions_mpc_previous_iter, dim_action=A)
            else:
                x0 = generate_mpc_action_init_random(len_horizon=H, dim_action=A)
            res = minimize(fun=self.compute_mean_lcb_trajectory, x0=x0, jac=True, args=(state_mu, state_var),
                           method="L-BFGS-B", bounds=self.actions_mapper.bounds, options=cc.actions_optimizer_params)
            if res.fun < opt_fun or (best is None and np.isnan(res.fun)):
                opt_fun, best = res.fun, res.x
        self.actions_mpc_previous_iter = best.copy()
        return self.actions_mapper.transform_action_mpc_to_action_model(best)

    def _random_shooting(self, state_mu, state_var):
        """optimize=False: the reference draws TWO sequences per restart and evaluates the second
        (:127-130 then :143); reproduced so that a seeded run sees the same candidates."""
        from ... import sharding
        cc = self.config.controller
        H, A = cc.len_horizon, self.actions_mapper.dim_action
        cands = []
        for idx_restart in range(cc.restarts_optim):
            if cc.init_from_previous_actions and self.actions_mpc_previous_iter is not None and idx_restart == 0:
                generate_mpc_action_init_frompreviousiter(self.actions_mpc_previous_iter, dim_action=A)
            else:
                generate_mpc_action_init_random(len_horizon=H, dim_action=A)
            cands.append(generate_mpc_action_init_random(len_horizon=H, dim_action=A))
        cands = np.stack(cands)
        B = cands.shape[0]
        world, rank = self._ranks()
        # Every rank draws the SAME B candidates (the reference's global numpy generator, seeded identically on all
        # ranks by the launcher; checked below through a checksum inside the records) and evaluates its contiguous slice; a
        # rank whose slice is empty (B < world) launches nothing and contributes an (inf, -1) record.
        lo, hi = sharding.shard_bounds(B, world, rank)
        eng = self.transition_model.engine
        failure = None
        try:
            out = self.evaluate_candidates(cands[lo:hi], state_mu, state_var, trajectories=True) if hi > lo else None
        except Exception as e:                   # noqa: BLE001 -- world > 1: handed on after the collective, see below
            if world == 1:
                raise
            # this rank must still enter the step's all_gather (its peers would wait in it until the watchdog fires): it
            # contributes an (inf, -1) record whose error flag every rank reads after the exchange
            failure, out = e, None
        local = torch.as_tensor(cands[lo:hi].reshape(hi - lo, H, A), device=eng.device)
        if failure is not None:
            local = local[:0]
        if world == 1:
            J, best, win = sharding.select_best_on_device(eng, out["J"], local, lo, B, group=sharding.LOCAL)
            self._cache_trajectory(out, B - 1)       # the reference caches the LAST evaluated trajectory, not the winner's (:279-283)
        else:
            # ... and so must every rank here (get_action reads the caches on all of them): the owner of the last
            # global candidate appends that trajectory to its record, so the step's whole exchange is ONE all_gather.
            D = self.transition_model.dim_state
            last_owner = sharding.owner_of(B - 1, B, world)
            if rank == last_owner and out is not None:
                extra = torch.cat([out[k][-1].reshape(-1) for k in ("mu", "Sig", "cost_mu", "cost_var")] + [out["J"][-1:]])
            else:
                extra = torch.zeros((H + 1) * (D + D * D + 2) + 1, dtype=F64, device=eng.device)
            # two trailing doubles: checksums of the state and of the drawn candidates (same on every rank, or the union of
            # the slices is not the single-GPU population)
            # three trailing doubles: an error flag, and checksums of the state and of the drawn candidates (same on every rank, or
            # the union of the slices is not the single-GPU population)
            sums = torch.tensor([0.0 if failure is None else 1.0, self._state_checksum(state_mu, state_var),
                                 float(np.dot(cands.ravel(), np.cos(np.arange(cands.size))))], dtype=F64, device=eng.device)
            J, best, win, extras = sharding.select_best_on_device(eng, None if out is None else out["J"], local, lo, B,
                                                                  extra=torch.cat([extra, sums]), group=self.process_group,
                                                                  raise_if_none=False)
            if failure is not None:
                raise failure
            failed = [r for r in range(world) if float(extras[r, -3]) != 0.0]
            if failed:
                raise RuntimeError(f"random shooting: the slice of rank(s) {failed} failed; no winner this step")
            self._assert_ranks_agree(extras[:, -2], "the state")
            self._assert_ranks_agree(extras[:, -1], "the drawn candidates")
            if win is None:
                raise FloatingPointError("no selectable candidate (all objectives NaN)")
            self._cache_packed_trajectory(extras[last_owner][:-3], H, D)
        self.best_candidate_index, self.best_candidate_J = best, J
        self.actions_mpc_previous_iter = win.numpy().reshape(-1).copy()
        return self.actions_mapper.transform_action_mpc_to_action_model(self.actions_mpc_previous_iter)

    def _cross_entropy_search(self, state_mu, state_var):
        """Batched replacement of the sequential scipy restarts: every iteration is ONE
        rollout launch over `cem_candidates` sequences drawn around the current mean (iteration 0: uniform,
        plus the shifted previous solution when init_from_previous_actions), the elites refit mean / std of a
        diagonal Gaussian over the optimiser vector in [0, 1]^(H*A); the best sequence ever evaluated wins."""
        cc = self.config.controller
        H, A = cc.len_horizon, self.actions_mapper.dim_action
        n, B = H * A, int(cc.cem_candidates)
        n_elite = max(2, int(round(B * cc.cem_elite_fraction)))
        rng = np.random                                        # the reference's global generator
        mean, std = np.full(n, 0.5), np.full(n, 0.5)
        best_x, best_J = None, np.inf
        for it in range(int(cc.cem_iterations)):
            if it == 0:
                cands = rng.uniform(0.0, 1.0, size=(B, n))
                if cc.init_from_previous_actions and self.actions_mpc_previous_iter is not None:
                    cands[0] = generate_mpc_action_init_frompreviousiter(self.actions_mpc_previous_iter, dim_action=A)
            else:
                cands = np.clip(mean + std * rng.standard_normal((B, n)), 0.0, 1.0)
                cands[0] = best_x                              # keep the incumbent
            out = self.evaluate_candidates(cands, state_mu, state_var, trajectories=True)
            J = out["J"].cpu().numpy()
            J = np.where(np.isnan(J), np.inf, J)
            order = np.argsort(J, kind="stable")
            if J[order[0]] < best_J:
                best_J, best_x = float(J[order[0]]), cands[order[0]].copy()
                self._cache_trajectory(out, int(order[0]))
            elites = cands[order[:n_elite]]
            mean, std = elites.mean(axis=0), elites.std(axis=0) + 1e-3
        if best_x is None:
            raise FloatingPointError("no finite objective among the candidates")
        self.best_candidate_J = best_J
        self.actions_mpc_previous_iter = best_x.copy()
        return self.actions_mapper.transform_action_mpc_to_action_model(best_x)

    def _device_cross_entropy_search(self, state_mu, state_var):
        """candidate_optimizer = "cem_device": the cross-entropy search of `_cross_entropy_search` with its whole loop
        on the GPU (gpmpc_cem_search: sampling, action mapper, rollout, elite selection and refit enqueued back to back,
        nothing read back between iterations).  Two host synchronisations per control step -- the winner, then its
        trajectory for the logging caches -- instead of one per iteration; the draws are Philox, keyed by a seed taken
        from numpy's global generator (so `np.random.seed` still makes a run reproducible)."""
        cc = self.config.controller
        H, A = cc.len_horizon, self.actions_mapper.dim_action
        B = int(cc.cem_candidates)
        n_elite = max(2, int(round(B * cc.cem_elite_fraction)))
        first = None
        if cc.init_from_previous_actions and self.actions_mpc_previous_iter is not None:
            first = generate_mpc_action_init_frompreviousiter(self.actions_mpc_previous_iter, dim_action=A)
        kw = {}
        if isinstance(self.actions_mapper, DerivativeActionMapper):
            kw = dict(max_change=np.asarray(self.config.actions.max_change_action_norm, dtype=np.float64),
                      action_prev=self.actions_mapper.action_model_previous_iter.numpy())
        seed = int(np.random.randint(0, 2 ** 62))          # the Philox key (np.random.seed keeps a run reproducible)
        tm = self.transition_model
        tm.set_cost(self.config.reward)
        world, rank = self._ranks()
        if world > 1:
            # candidates sharded over the ranks: each draws its slice from the shared Philox key, one elite merge per iteration.
            # The key, the state and the starting vector are rank 0's on every rank (ONE small broadcast per control step: the
            # union of the slices is the single-GPU population only if they are identical everywhere; nothing but the launcher
            # enforced that before).
            from ... import sharding
            n = H * A
            head = np.concatenate([[float(seed >> 31), float(seed & 0x7fffffff), 1.0 if first is not None else 0.0],
                                   np.zeros(n) if first is None else np.asarray(first, dtype=np.float64).ravel(),
                                   np.asarray(state_mu, dtype=np.float64).ravel(), np.asarray(state_var, dtype=np.float64).ravel()])
            mine = self._state_checksum(state_mu, state_var)
            head = sharding.broadcast_from_first(head, tm.engine.device, self.process_group)
            # the ranks must sit at the SAME state (the other searches verify it inside their exchange): compared with rank 0's
            # here, and the verdict is taken by all ranks together (one rank raising alone would leave the others in the search's
            # collectives)
            theirs = self._state_checksum(head[3 + n:3 + n + np.asarray(state_mu).size], head[3 + n + np.asarray(state_mu).size:])
            if sharding.any_rank(mine != theirs, tm.engine.device, self.process_group):
                raise RuntimeError("candidate sharding: the ranks disagree on the state; every rank must see the same observation "
                                   "(or switch ControllerConfig.shard_over_ranks off)")
            seed = (int(head[0]) << 31) | int(head[1])
            first = head[3:3 + n].copy() if head[2] != 0.0 else None
            nmu = np.asarray(state_mu).size
            state_mu = head[3 + n:3 + n + nmu].reshape(np.shape(state_mu))
            state_var = head[3 + n + nmu:].reshape(np.shape(state_var))
            best_x, best_J = sharding.sharded_cem_search(
                tm.engine, np.asarray(state_mu, dtype=np.float64), np.asarray(state_var, dtype=np.float64), B, H, A,
                int(cc.cem_iterations), n_elite, seed=seed, include_time=tm.config.include_time_model,
                time0=float(self.iter_ctrl), first_candidate=first, group=self.process_group, **kw)
        else:
            best_x, best_J = tm.engine.cem_search(
                np.asarray(state_mu, dtype=np.float64), np.asarray(state_var, dtype=np.float64), B, H, A,
                int(cc.cem_iterations), n_elite, seed=seed, include_time=tm.config.include_time_model,
                time0=float(self.iter_ctrl), first_candidate=first, **kw)
        if not np.isfinite(best_J):
            raise FloatingPointError("no finite objective among the candidates")
        self.num_rollouts += B * int(cc.cem_iterations)
        out = self.evaluate_candidates(best_x[None], state_mu, state_var, trajectories=True)
        self._cache_trajectory(out, 0)
        self.best_candidate_J = best_J
        self.actions_mpc_previous_iter = best_x.copy()
        return self.actions_mapper.transform_action_mpc_to_action_model(best_x)

    def objective_and_gradient_batch(self, actions_mpc_batch, obs_mu, obs_var):
        """(B, H*A) optimiser vectors -> (J (B,), dJ/d(actions_mpc) (B, H*A)) in one gpmpc_rollout_grad launch."""
        X = np.asarray(actions_mpc_batch, dtype=np.float64)
        acts = self.actions_mapper.mpc_to_model_batch(X)
        self.transition_model.set_cost(self.config.reward)
        if self.analytic_gradient:
            try:
                out = self.transition_model.objective_and_gradient_batch(acts, obs_mu, obs_var, self.iter_ctrl)
            except GpmpcError as e:
                if e.code != GPMPC_ERR_LIMIT:
                    raise
                self.analytic_gradient = False         # shape outside the gradient kernels: difference the rollout
            else:
                self.num_rollouts += X.shape[0]
                host = self.transition_model.engine.host_views(out)
                return host["J"].numpy(), self.actions_mapper.chain_grad_model_to_mpc_batch(host["grad"].numpy())
        J, g_model, _ = self._objective_and_gradient_by_differences(acts, obs_mu, obs_var)
        return J, self.actions_mapper.chain_grad_model_to_mpc_batch(g_model)

    def _batched_lbfgs_search(self, state_mu, state_var):
        """All restarts at once.  The reference runs `restarts_optim` scipy L-BFGS-B solves one after
        the other, every function evaluation a forward + autograd backward of ONE sequence (:125-141).  Here the same
        scipy solver runs once per restart, each in its own thread, but the threads advance in lockstep: whenever every
        unfinished restart is waiting for an evaluation, ONE objective + analytic-gradient launch serves them all.  Each
        restart therefore sees exactly the values the sequential loop would hand it (the kernels are bitwise independent
        of the batch composition), ends at the same point, and the keep-the-best rule (:146-148) picks the same winner;
        the wall time is that of the longest restart instead of their sum."""
        import threading
        cc = self.config.controller
        H, A = cc.len_horizon, self.actions_mapper.dim_action
        B = int(cc.lbfgs_candidates or cc.restarts_optim)
        x0s = []
        for b in range(B):
            if cc.init_from_previous_actions and self.actions_mpc_previous_iter is not None and b == 0:
                x0s.append(generate_mpc_action_init_frompreviousiter(self.actions_mpc_previous_iter, dim_action=A))
            else:
                x0s.append(generate_mpc_action_init_random(len_horizon=H, dim_action=A))
        # restarts sharded over the ranks (one process per GPU): every rank drew the SAME B starting points above (numpy's
        # global generator, seeded identically by the launcher) and solves its contiguous slice; the winners meet in ONE
        # all_gather of [fun, restart index, solution] below
        from ... import sharding
        world, rank = self._ranks()
        if world > 1:
            # rank 0's starting points on every rank (one small broadcast per control step instead of trusting the launcher's seeds)
            x0s = list(sharding.broadcast_from_first(np.stack(x0s), self.transition_model.engine.device, self.process_group))
        r_lo, r_hi = sharding.shard_bounds(B, world, rank)
        B_all, x0s, B = B, x0s[r_lo:r_hi], r_hi - r_lo
        cond = threading.Condition()
        pending, results, state = {}, {}, {"running": B, "launches": 0, "error": None}

        def flush():                                   # called with the lock held, by the last thread to arrive
            idx = sorted(pending)
            J, G = np.full(len(idx), np.nan), np.zeros((len(idx), H * A))
            try:
                J, G = self.objective_and_gradient_batch(np.stack([pending[i] for i in idx]), state_mu, state_var)
            except BaseException as e:                 # every waiting restart is served -- nobody may wait forever ...
                state["error"] = e if isinstance(e, Exception) else RuntimeError(repr(e))
                if not isinstance(e, Exception):       # ... and KeyboardInterrupt / SystemExit go on in the flushing thread
                    raise
            finally:
                state["launches"] += 1
                for k, i in enumerate(idx):
                    results[i] = (float(J[k]), G[k].copy())
                pending.clear()
                cond.notify_all()

        def make_fun(i):
            def fun(x):
                with cond:
                    pending[i] = np.array(x, dtype=np.float64)
                    if len(pending) == state["running"]:
                        flush()
                    waited = 0.0
                    while i not in results:
                        # a rendezvous that never completes (a sibling died outside `worker`'s bookkeeping) must not hang the step
                        if not cond.wait(timeout=5.0):
                            waited += 5.0
                            if waited >= 600.0:
                                pending.pop(i, None)
                                raise TimeoutError(f"lockstep evaluation of restart {i} not served within 600 s")
                    out = results.pop(i)
                if state["error"] is not None:
                    raise RuntimeError("batched evaluation failed") from state["error"]
                return out
            return fun

        solved = [None] * B

        def worker(i):
            try:
                solved[i] = minimize(fun=make_fun(i), x0=x0s[i], jac=True, method="L-BFGS-B",
                                     bounds=self.actions_mapper.bounds, options=cc.actions_optimizer_params)
            except BaseException as e:
                solved[i] = e
            finally:
                with cond:
                    state["running"] -= 1
                    if pending and len(pending) == state["running"]:
                        flush()

        threads = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(B)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        failure = next((r for r in solved if isinstance(r, BaseException)), None)
        if failure is not None and world == 1:
            raise failure
        # world > 1: a rank whose solves failed must still enter the collective below (the others would wait in it until the
        # watchdog fires); its record carries an error flag and every rank raises after the gather
        self.lbfgs_evaluations = state["launches"]
        self.candidates_final_J = None if failure is not None else np.array([r.fun for r in solved])
        opt_fun, best, best_i = np.inf, None, -1
        for i, res in enumerate(solved if failure is None else []):     # the reference's keep-the-best rule, in restart order
            if res.fun < opt_fun or (best is None and np.isnan(res.fun)):
                opt_fun, best, best_i = res.fun, res.x, r_lo + i
        if world > 1:
            # a NaN adopted by a rank whose slice does not start at restart 0 must not win (reference rule: only the FIRST
            # restart's NaN is adopted) -- such a slice contributes its best finite result, or nothing
            if best is not None and np.isnan(opt_fun) and best_i != 0:
                opt_fun, best, best_i = np.inf, None, -1
                for i, res in enumerate(solved):
                    if res.fun < opt_fun:
                        opt_fun, best, best_i = res.fun, res.x, r_lo + i
            n = H * A
            rec = torch.zeros(2 + n + 2, dtype=F64)     # [fun | restart | solution | error flag | state checksum]
            rec[0], rec[1] = (float(opt_fun), float(best_i)) if best is not None else (float("inf"), -1.0)
            if best is not None:
                rec[2:2 + n] = torch.as_tensor(np.asarray(best, dtype=np.float64))
            rec[2 + n] = 0.0 if failure is None else 1.0
            rec[3 + n] = self._state_checksum(state_mu, state_var)
            dev = self.transition_model.engine.device
            _, flat = sharding._gather_records(rec.to(dev), self.process_group)
            host = flat.cpu().view(world, 4 + n)
            if failure is not None:
                raise failure
            failed = [r for r in range(world) if float(host[r, 2 + n]) != 0.0]
            if failed:
                raise RuntimeError(f"lockstep L-BFGS restarts: the solves of rank(s) {failed} failed; no winner this step")
            self._assert_ranks_agree(host[:, 3 + n], "the state")
            opt_fun, best_i, win = sharding._winner_of(host, world, n, 1)
            best = win.numpy().reshape(-1).copy()
            self.candidates_final_J = None             # only this rank's slice was solved here
        self.best_candidate_index, self.best_candidate_J = best_i, float(opt_fun)
        out = self.evaluate_candidates(best[None], state_mu, state_var, trajectories=True)
        self._cache_trajectory(out, 0)
        self.actions_mpc_previous_iter = best.copy()
        return self.actions_mapper.transform_action_mpc_to_action_model(best)

    def _get_random_actions(self, state_mu, state_var):
        """Reference :155-163: one random sequence, evaluated only to fill the logging caches."""
        H, A = self.config.controller.len_horizon, self.actions_mapper.dim_action
        actions_mpc = generate_mpc_action_init_random(len_horizon=H, dim_action=A)
        self._prepare()
        out = self.evaluate_candidates(actions_mpc[None], state_mu, state_var, trajectories=True)
        self._cache_trajectory(out, 0)
        return self.actions_mapper.transform_action_mpc_to_action_model(actions_mpc)
