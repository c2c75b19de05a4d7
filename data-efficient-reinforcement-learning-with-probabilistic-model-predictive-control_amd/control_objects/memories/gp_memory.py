"""Replay buffer + thresholded admission into the GP memory (reference
control_objects/memories/gp_memory.py).  Host bookkeeping, O(1) per env step.  The reference's
buffer-growth path raises TypeError once points_batch_memory is exceeded (torch.cat misuse,
gp_memory.py:35-40,70-71); here the buffers really grow."""
import numpy as np
import torch

from ..utils.data_utils import form_model_input

F64 = torch.float64


class Memory:
    def __init__(self, config, dim_input, dim_state, include_time_model=False, step_model=1):
        self.config = config
        self.include_time_model = include_time_model
        self.dim_input = dim_input
        self.dim_state = dim_state
        self.step_model = step_model
        n = config.points_batch_memory
        self.inputs = torch.empty((n, dim_input), dtype=F64)
        self.states_next = torch.empty((n, dim_state), dtype=F64)
        self.rewards = torch.empty(n, dtype=F64)
        self.iter_ctrls = torch.empty(n, dtype=F64)
        self.errors = torch.empty((n, dim_state), dtype=F64)
        self.stds = torch.empty((n, dim_state), dtype=F64)
        self.model_inputs = torch.empty((n, dim_input), dtype=F64)
        self.model_targets = torch.empty((n, dim_state), dtype=F64)
        self.active_data_mask = np.empty(n, dtype=bool)
        self.len_mem = 0
        self.len_mem_last_processed = 0
        self.len_mem_model = 0

    @staticmethod
    def _grown(t, extra):
        return torch.cat((t, torch.empty((extra,) + tuple(t.shape[1:]), dtype=t.dtype)))

    def _grow_raw(self):
        n = self.config.points_batch_memory
        for name in ("inputs", "states_next", "rewards", "iter_ctrls", "errors", "stds"):
            setattr(self, name, self._grown(getattr(self, name), n))
        self.active_data_mask = np.concatenate((self.active_data_mask, np.empty(n, dtype=bool)))

    def add(self, state, action_model, state_next, reward, iter_ctrl=0, **kwargs):
        if len(self.inputs) < self.len_mem + 1:
            self._grow_raw()
        i = self.len_mem
        self.inputs[i] = form_model_input(state, action_model, iter_ctrl, self.include_time_model, self.dim_input)
        self.states_next[i] = state_next
        self.rewards[i] = reward
        self.iter_ctrls[i] = iter_ctrl
        keep = True
        if self.config.check_errors_for_storage:
            pred = kwargs.get("predicted_state")
            if pred is not None:
                err = torch.abs(torch.as_tensor(np.asarray(pred), dtype=F64) - state_next)
                keep = bool(torch.any(err > self.config.min_error_prediction_state_for_memory))
                self.errors[i] = err
            else:
                self.errors[i] = float("nan")
            std = kwargs.get("predicted_state_std")
            if std is not None:
                std = torch.as_tensor(np.asarray(std), dtype=F64)
                keep = keep and bool(torch.any(std > self.config.min_prediction_state_std_for_memory))
                self.stds[i] = std
            else:
                self.stds[i] = float("nan")
        self.active_data_mask[i] = keep
        self.len_mem += 1

    def get_indexes_to_process(self):
        return np.arange(self.len_mem_last_processed, self.len_mem, self.step_model)

    def get_indexes_processed(self):
        return np.arange(0, self.len_mem_last_processed, self.step_model)

    def get_memory_by_index(self, indexes):
        inputs = self.inputs[indexes]
        targets = self.states_next[indexes + self.step_model - 1] - self.inputs[indexes, :self.dim_state]
        return inputs, targets

    def prepare_for_model(self):
        idx = self.get_indexes_to_process()
        idx = idx[self.active_data_mask[idx]]
        n_new = len(idx)
        while len(self.model_inputs) < self.len_mem_model + n_new + 1:
            self.model_inputs = self._grown(self.model_inputs, self.config.points_batch_memory)
            self.model_targets = self._grown(self.model_targets, self.config.points_batch_memory)
        x, y = self.get_memory_by_index(idx)
        self.model_inputs[self.len_mem_model:self.len_mem_model + n_new] = x
        self.model_targets[self.len_mem_model:self.len_mem_model + n_new] = y
        self.len_mem_model += n_new
        self.len_mem_last_processed = self.len_mem

    def get_memory_total(self):
        return self.get_memory_by_index(self.get_indexes_processed())

    def get_mask_model_inputs(self):
        return self.active_data_mask[self.get_indexes_processed()]

    def get(self):
        if self.len_mem_model > 0:
            return self.model_inputs[:self.len_mem_model], self.model_targets[:self.len_mem_model]
        # empty memory: one all-zero dummy point so that N = 1, beta = 0 (reference :109-111)
        return torch.zeros((1, self.dim_input), dtype=F64), torch.zeros((1, self.dim_state), dtype=F64)
