"""Plain-Python configuration objects with the reference's class names, constructor keywords
and attribute names (rl_gp_mpc/config_classes/*.py), so the reference's example config files
(examples/*/config_*.py) build the same objects against this package.

Differences, on purpose: tensors are created with an explicit dtype=float64 instead of flipping
torch's global default dtype at import time (reference: total_config.py:11), and mutable default
arguments are copied.
"""
import copy

import torch

F64 = torch.float64


def _t(v):
    return torch.as_tensor(v, dtype=F64) if isinstance(v, (list, tuple)) else v


def _tensorise_lists(obj):
    for k, v in list(vars(obj).items()):
        if isinstance(v, (list, tuple)):
            setattr(obj, k, torch.as_tensor(v, dtype=F64))


class ObservationConfig:
    def __init__(self, obs_var_norm=(1e-6, 1e-6, 1e-6)):
        # diagonal covariance of the normalised observation (reference observation_config.py:11)
        self.obs_var_norm = torch.diag(torch.as_tensor(list(obs_var_norm), dtype=F64))


class RewardConfig:
    def __init__(self, target_state_norm=(1, 0.5, 0.5), weight_state=(1, 0.1, 0.1), weight_state_terminal=(10, 5, 5),
                 target_action_norm=(0.5,), weight_action=(0.05,), exploration_factor=3, use_constraints=False,
                 state_min=(-0.1, 0.05, 0.05), state_max=(1.1, 0.95, 0.925), area_multiplier=1,
                 clip_lower_bound_cost_to_0=False):
        self.target_state_norm = list(target_state_norm)
        self.weight_state = list(weight_state)
        self.weight_state_terminal = list(weight_state_terminal)
        self.target_action_norm = list(target_action_norm)
        self.weight_action = list(weight_action)
        self.exploration_factor = exploration_factor
        self.use_constraints = use_constraints
        self.state_min = list(state_min)
        self.state_max = list(state_max)
        self.area_multiplier = area_multiplier
        self.clip_lower_bound_cost_to_0 = clip_lower_bound_cost_to_0
        _tensorise_lists(self)
        # derived (reference reward_config.py:54-64)
        self.weight_matrix_cost = torch.block_diag(torch.diag(self.weight_state), torch.diag(self.weight_action))
        self.weight_matrix_cost_terminal = torch.diag(self.weight_state_terminal)
        self.target_state_action_norm = torch.cat((self.target_state_norm, self.target_action_norm))


class ActionsConfig:
    def __init__(self, limit_action_change=False, max_change_action_norm=(0.05,)):
        self.limit_action_change = limit_action_change
        self.max_change_action_norm = list(max_change_action_norm)
        _tensorise_lists(self)


class MemoryConfig:
    def __init__(self, check_errors_for_storage=True, min_error_prediction_state_for_memory=(3e-4, 3e-4, 3e-4),
                 min_prediction_state_std_for_memory=(3e-3, 3e-3, 3e-3), points_batch_memory=1500):
        self.check_errors_for_storage = check_errors_for_storage
        self.min_error_prediction_state_for_memory = list(min_error_prediction_state_for_memory)
        self.min_prediction_state_std_for_memory = list(min_prediction_state_std_for_memory)
        self.points_batch_memory = points_batch_memory
        _tensorise_lists(self)


class TrainingConfig:
    def __init__(self, lr_train=7e-3, iter_train=15, training_frequency=25, clip_grad_value=1e-3, print_train=False,
                 step_print_train=5, device="auto"):
        self.lr_train = lr_train
        self.iter_train = iter_train
        self.training_frequency = training_frequency
        self.clip_grad_value = clip_grad_value
        self.print_train = print_train
        self.step_print_train = step_print_train
        # where the training loss is evaluated: "auto" / "hip" = gpmpc_mll on the GPU of the training process.  There is no
        # CPU expression of the loss in this package (the reference's runs in gpytorch): anything else is refused HERE, in
        # the parent, instead of failing inside the spawned child
        if device not in ("auto", "hip"):
            raise ValueError(f"TrainingConfig.device={device!r}: the training loss runs on the GPU only ('auto' or 'hip')")
        self.device = device


_DEFAULT_OPTIMIZER = {"disp": None, "maxcor": 30, "ftol": 1e-99, "gtol": 1e-99, "eps": 1e-2, "maxfun": 30,
                      "maxiter": 30, "iprint": -1, "maxls": 30, "finite_diff_rel_step": None}


class ControllerConfig:
    """Reference keywords (controller_config.py:1-25) plus optional ones for the batched optimisers that replace
    the sequential scipy restarts: `candidate_optimizer="cem"` (cross-entropy search, one
    rollout launch per iteration) or `"lbfgs"` (the reference's scipy L-BFGS-B restarts advanced in lockstep: one
    objective + analytic-gradient launch serves every restart's pending evaluation; same result as the sequential
    loop, wall time of the longest restart)."""

    def __init__(self, len_horizon=15, actions_optimizer_params=None, init_from_previous_actions=True,
                 restarts_optim=1, optimize=True, num_repeat_actions=1,
                 candidate_optimizer=None, cem_candidates=256, cem_iterations=4, cem_elite_fraction=0.1,
                 lbfgs_candidates=None, shard_over_ranks=False):
        self.len_horizon = len_horizon
        self.actions_optimizer_params = dict(_DEFAULT_OPTIMIZER if actions_optimizer_params is None
                                             else actions_optimizer_params)
        self.init_from_previous_actions = init_from_previous_actions
        self.restarts_optim = restarts_optim      # with optimize=False: number of random candidates (one GPU launch)
        self.optimize = optimize
        self.num_repeat_actions = num_repeat_actions
        self.candidate_optimizer = candidate_optimizer      # None: scipy L-BFGS-B (reference behaviour); "cem"
        self.cem_candidates = cem_candidates                # candidates per iteration = one kernel launch
        self.cem_iterations = cem_iterations
        self.cem_elite_fraction = cem_elite_fraction
        self.lbfgs_candidates = lbfgs_candidates            # None: restarts_optim starting points
        # one process per GPU (torch.distributed initialised by the launcher): split the candidates / restarts over the ranks.
        # Opt-in: every rank must then call get_action with the same observation (checked inside the exchange)
        self.shard_over_ranks = shard_over_ranks


def _broadcast(v, shape):
    """Scalar / per-model vector -> tensor of `shape` (reference functions_process_config.py:29-36)."""
    t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=F64)
    t = t.to(F64)
    if t.ndim < len(shape):
        t = t.unsqueeze(-1)
    return t * torch.ones(shape, dtype=F64)


def _with_time_column(ls, ls_time, num_models, num_inputs):
    """Lengthscale table whose last column is the time lengthscale (functions_process_config.py:18-26)."""
    out = torch.empty((num_models, num_inputs), dtype=F64)
    t = ls if isinstance(ls, torch.Tensor) else torch.as_tensor(ls, dtype=F64)
    if t.ndim == 1:
        out[:, :-1] = t[:, None].expand(num_models, num_inputs - 1)
    else:
        out[:, :-1] = t
    out[:, -1] = ls_time
    return out


class ModelConfig:
    def __init__(self, gp_init=None, init_lengthscale_time=100, min_std_noise=1e-3, max_std_noise=3e-1,
                 min_outputscale=1e-5, max_outputscale=0.95, min_lengthscale=4e-3, max_lengthscale=25.0,
                 min_lengthscale_time=10, max_lengthscale_time=10000, include_time_model=False):
        if gp_init is None:
            gp_init = {"noise_covar.noise": [1e-4] * 3, "base_kernel.lengthscale": [[0.75] * 4] * 3,
                       "outputscale": [5e-2] * 3}
        self.include_time_model = include_time_model
        self.min_std_noise = min_std_noise
        self.max_std_noise = max_std_noise
        self.min_outputscale = min_outputscale
        self.max_outputscale = max_outputscale
        self.min_lengthscale = min_lengthscale
        self.max_lengthscale = max_lengthscale
        self.min_lengthscale_time = min_lengthscale_time
        self.max_lengthscale_time = max_lengthscale_time
        self.init_lengthscale_time = init_lengthscale_time
        self.gp_init = {k: _t(copy.deepcopy(v)) for k, v in gp_init.items()}

    def extend_dimensions_params(self, dim_state, dim_input):
        """Per-GP broadcasting of scalar settings (reference model_config.py:46-67)."""
        for name in ("min_std_noise", "max_std_noise", "min_outputscale", "max_outputscale"):
            setattr(self, name, _broadcast(getattr(self, name), (dim_state,)))
        self.gp_init["noise_covar.noise"] = _broadcast(self.gp_init["noise_covar.noise"], (dim_state,))
        self.gp_init["outputscale"] = _broadcast(self.gp_init["outputscale"], (dim_state,))
        if self.include_time_model:
            self.min_lengthscale = _with_time_column(self.min_lengthscale, self.min_lengthscale_time, dim_state, dim_input)
            self.max_lengthscale = _with_time_column(self.max_lengthscale, self.max_lengthscale_time, dim_state, dim_input)
            self.gp_init["base_kernel.lengthscale"] = _with_time_column(
                self.gp_init["base_kernel.lengthscale"], self.init_lengthscale_time, dim_state, dim_input)
        else:
            self.min_lengthscale = _broadcast(self.min_lengthscale, (dim_state, dim_input))
            self.max_lengthscale = _broadcast(self.max_lengthscale, (dim_state, dim_input))
            self.gp_init["base_kernel.lengthscale"] = _broadcast(self.gp_init["base_kernel.lengthscale"],
                                                                 (dim_state, dim_input))


class VisuConfig:
    """Accepted for call compatibility (reference visu_config.py); visualisation is out of scope."""

    def __init__(self, save_render_env=True, render_live_plot_2d=True, render_env=True, save_live_plot_2d=False):
        self.save_render_env = save_render_env
        self.render_live_plot_2d = render_live_plot_2d
        self.render_env = render_env
        self.save_live_plot_2d = save_live_plot_2d


class Config:
    def __init__(self, observation_config=None, reward_config=None, actions_config=None, model_config=None,
                 memory_config=None, training_config=None, controller_config=None):
        self.observation = observation_config or ObservationConfig()
        self.reward = reward_config or RewardConfig()
        self.actions = actions_config or ActionsConfig()
        self.model = model_config or ModelConfig()
        self.memory = memory_config or MemoryConfig()
        self.training = training_config or TrainingConfig()
        self.controller = controller_config or ControllerConfig()
