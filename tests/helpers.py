"""Shared test helpers (fixtures -> oracle objects, error metrics)."""
import os

import numpy as np

from oracle import synth
from oracle.gpmpc_oracle import Factors

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def workload_of(g):
    return synth.Workload(X=g["X"], Y=g["Y"], lengthscales=g["lengthscales"], outputscales=g["outputscales"],
                          noises=g["noises"], actions=g["actions"], mu0=g["mu0"], S0=g["S0"],
                          include_time=bool(g["include_time"]), time0=float(g["time0"]), target=g["target"],
                          W=g["W"], W_T=g["W_T"], kappa=float(g["kappa"]))


def factors_of(w):
    return Factors(w.X, w.Y, w.lengthscales, w.outputscales, w.noises)


def rel_err(a, b):
    """max |a-b| / max|b|  (scale-relative, robust to entries that pass through zero)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


_REPORT = {}


def record(test, **values):
    """Keep the ACHIEVED error of a parity test (not just pass / fail): collected into gpurun_out/parity_report.json
    at interpreter exit; the GPU run's copy is committed under profiles/."""
    import atexit
    import json
    if not _REPORT:
        def dump():
            out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
            try:
                os.makedirs(out, exist_ok=True)
                with open(os.path.join(out, "parity_report.json"), "w") as fh:
                    json.dump(_REPORT, fh, indent=1, sort_keys=True)
            except OSError:
                pass
        atexit.register(dump)
    _REPORT.setdefault(test, {}).update({k: float(v) for k, v in values.items()})


def make_controller(w, limit_action_change=False, optimize=False, restarts=1, clip=False, engine=None,
                    optimizer_params=None, shard=True):
    """A GpMpcController of THIS package configured like the golden generator configured the
    reference's (tools/gen_golden.py make_ref_controller) and fed workload `w` as GP memory."""
    import torch
    import gp_mpc_amd  # noqa: F401
    from gp_mpc_amd.config_classes import (Config, ControllerConfig, ActionsConfig, RewardConfig, ObservationConfig,
                                           MemoryConfig, ModelConfig, TrainingConfig)
    from gp_mpc_amd import GpMpcController
    N, D, A, E, H, B = w.dims
    reward = RewardConfig(target_state_norm=list(w.target[:D]), weight_state=list(np.diag(w.W)[:D]),
                          weight_state_terminal=list(np.diag(w.W_T)), target_action_norm=list(w.target[D:]),
                          weight_action=list(np.diag(w.W)[D:]), exploration_factor=w.kappa,
                          clip_lower_bound_cost_to_0=clip)
    model = ModelConfig(gp_init={"noise_covar.noise": list(w.noises), "base_kernel.lengthscale": w.lengthscales.tolist(),
                                 "outputscale": list(w.outputscales)}, include_time_model=w.include_time)
    if w.include_time:                      # the time column of the lengthscales comes from the workload
        model.init_lengthscale_time = 0.0
    ctrl_cfg = ControllerConfig(len_horizon=H, restarts_optim=restarts, optimize=optimize,
                                actions_optimizer_params=optimizer_params, shard_over_ranks=shard)
    cfg = Config(observation_config=ObservationConfig(obs_var_norm=list(np.diag(w.S0))), reward_config=reward,
                 actions_config=ActionsConfig(limit_action_change=limit_action_change, max_change_action_norm=[0.3] * A),
                 model_config=model, memory_config=MemoryConfig(points_batch_memory=max(16, N + 8)),
                 training_config=TrainingConfig(training_frequency=10 ** 9), controller_config=ctrl_cfg)
    c = GpMpcController(np.zeros(D), np.ones(D), np.zeros(A), np.ones(A), cfg, engine=engine)
    if w.include_time:
        for a, m in enumerate(c.transition_model.models):
            m.initialize(**{"covar_module.base_kernel.lengthscale": w.lengthscales[a]})
    c.memory.model_inputs[:N] = torch.as_tensor(w.X)
    c.memory.model_targets[:N] = torch.as_tensor(w.Y)
    c.memory.len_mem_model = N
    c.iter_ctrl = int(w.time0)
    return c
