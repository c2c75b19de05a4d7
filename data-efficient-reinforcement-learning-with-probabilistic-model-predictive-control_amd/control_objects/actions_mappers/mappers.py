"""Optimiser vector in [0,1]^(H*A)  <->  model actions (H, A)  <->  raw env actions.

Reference: control_objects/actions_mappers/{abstract,normalization,derivative}_action_mapper.py.
The mpc->model transforms also exist in a batched numpy form (`*_batch`) because the GPU path
evaluates many candidate sequences per launch.
"""
import numpy as np
import torch

F64 = torch.float64


class AbstractActionMapper:
    def __init__(self, action_low, action_high, len_horizon, config):
        self.config = config
        self.action_low = torch.as_tensor(np.asarray(action_low), dtype=F64)
        self.action_high = torch.as_tensor(np.asarray(action_high), dtype=F64)
        self.dim_action = len(action_low)
        self.len_horizon = len_horizon
        self.n_iter_ctrl = 0
        self.bounds = [(0, 1)] * self.dim_action * len_horizon

    def norm_action(self, action):
        return (torch.as_tensor(np.asarray(action), dtype=F64) - self.action_low) / (self.action_high - self.action_low)

    def denorm_action(self, normed_action, update_internals=False):
        if update_internals:           # the action is about to be applied to the environment
            self.n_iter_ctrl += 1
        return torch.as_tensor(np.asarray(normed_action), dtype=F64) * (self.action_high - self.action_low) + self.action_low

    def transform_action_raw_to_action_model(self, action_raw):
        return self.norm_action(action_raw)

    def transform_action_model_to_action_raw(self, action_model, update_internals=False):
        return self.denorm_action(action_model, update_internals=update_internals)

    def transform_action_mpc_to_action_model(self, action_mpc):
        raise NotImplementedError

    def transform_action_mpc_to_action_raw(self, action_mpc, update_internals=False):
        return self.transform_action_model_to_action_raw(self.transform_action_mpc_to_action_model(action_mpc),
                                                         update_internals=update_internals)

    # batched (B, H*A) -> (B, H, A), numpy
    def mpc_to_model_batch(self, actions_mpc):
        raise NotImplementedError

    def chain_grad_model_to_mpc(self, grad_model):
        """dJ/d(action_model) (H, A) -> dJ/d(action_mpc) (H*A,) with pass-through clamps."""
        raise NotImplementedError

    def chain_grad_model_to_mpc_batch(self, grad_model):
        """(B, H, A) -> (B, H*A)."""
        return np.stack([self.chain_grad_model_to_mpc(g) for g in np.asarray(grad_model, dtype=np.float64)])


class NormalizationActionMapper(AbstractActionMapper):
    """Identity reshape (reference normalization_action_mapper.py:21-23)."""

    def transform_action_mpc_to_action_model(self, action_mpc):
        a = torch.as_tensor(np.asarray(action_mpc), dtype=F64)
        return torch.atleast_2d(a.reshape(self.len_horizon, -1))

    def mpc_to_model_batch(self, actions_mpc):
        a = np.asarray(actions_mpc, dtype=np.float64)
        return a.reshape(a.shape[0], self.len_horizon, self.dim_action)

    def chain_grad_model_to_mpc(self, grad_model):
        return np.asarray(grad_model, dtype=np.float64).reshape(-1)


class DerivativeActionMapper(AbstractActionMapper):
    """Scaled deltas + cumulative sum + clamp to [0,1] whose backward is the identity
    (reference derivative_action_mapper.py:28-35, utils/pytorch_utils.py:4-13)."""

    def __init__(self, action_low, action_high, len_horizon, config):
        super().__init__(action_low, action_high, len_horizon, config)
        self.action_model_previous_iter = torch.rand(self.dim_action, dtype=F64)

    def transform_action_model_to_action_raw(self, action_model, update_internals=False):
        if update_internals:
            self.action_model_previous_iter = torch.as_tensor(np.asarray(action_model[0]), dtype=F64)
        return self.denorm_action(action_model, update_internals=update_internals)

    def _max_change(self):
        return np.asarray(self.config.max_change_action_norm, dtype=np.float64)

    def mpc_to_model_batch(self, actions_mpc):
        a = np.asarray(actions_mpc, dtype=np.float64)
        a = a.reshape(a.shape[0], self.len_horizon, self.dim_action)
        m = self._max_change()
        d = a * 2.0 * m - m
        d[:, 0] += self.action_model_previous_iter.numpy()
        return np.clip(np.cumsum(d, axis=1), 0.0, 1.0)

    def transform_action_mpc_to_action_model(self, action_mpc):
        a = np.asarray(action_mpc, dtype=np.float64).reshape(1, -1)
        return torch.as_tensor(self.mpc_to_model_batch(a)[0], dtype=F64)

    def chain_grad_model_to_mpc(self, grad_model):
        # model[t] = clamp(sum_{s<=t} (2 m u[s] - m) + prev), clamp backward = identity:
        # dJ/du[s] = 2 m * sum_{t>=s} dJ/dmodel[t]
        g = np.asarray(grad_model, dtype=np.float64)
        tail = np.cumsum(g[::-1], axis=0)[::-1]
        return (2.0 * self._max_change() * tail).reshape(-1)
