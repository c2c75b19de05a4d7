"""Observation -> normalised state in [0,1] (reference
control_objects/observations_states_mappers/{abstract,normalization}_observation_state_mapper.py)."""
import numpy as np
import torch

F64 = torch.float64


class AbstractObservationStateMapper:
    def __init__(self, observation_low, observation_high, config):
        self.config = config
        self.obs_low = torch.as_tensor(np.asarray(observation_low), dtype=F64)
        self.obs_high = torch.as_tensor(np.asarray(observation_high), dtype=F64)
        self.var_norm_factor = (self.obs_high - self.obs_low) ** 2
        self.dim_observation = len(observation_low)
        self.dim_state = self.dim_observation

    def get_state(self, obs, obs_var, update_internals):
        raise NotImplementedError


class NormalizationObservationStateMapper(AbstractObservationStateMapper):
    def get_state(self, obs, obs_var=None, update_internals=False):
        state = (torch.as_tensor(np.asarray(obs), dtype=F64) - self.obs_low) / (self.obs_high - self.obs_low)
        if obs_var is None:
            return state, self.config.obs_var_norm
        return state, torch.as_tensor(np.asarray(obs_var), dtype=F64) / self.var_norm_factor
