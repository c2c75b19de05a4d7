"""Per-control-step log record; field set = reference controllers/iteration_info_class.py:7-34."""
import numpy as np
import torch

NUM_DECIMALS_REPR = 3


class IterationInformation:
    FIELDS = ("iteration", "state", "cost", "cost_std", "mean_predicted_cost", "mean_predicted_cost_std",
              "lower_bound_mean_predicted_cost", "predicted_idxs", "predicted_states", "predicted_states_std",
              "predicted_actions", "predicted_costs", "predicted_costs_std")

    def __init__(self, **kw):
        missing = [f for f in self.FIELDS if f not in kw]
        if missing or len(kw) != len(self.FIELDS):
            raise TypeError(f"IterationInformation needs exactly {self.FIELDS}; missing {missing}")
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    def to_arrays(self):
        for k, v in vars(self).items():
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.detach().cpu().numpy())

    def to_tensors(self):
        for k, v in vars(self).items():
            if isinstance(v, np.ndarray):
                setattr(self, k, torch.as_tensor(v))

    def __str__(self):
        np.set_printoptions(precision=NUM_DECIMALS_REPR, suppress=True)
        lines = [""]
        for k, v in vars(self).items():
            if isinstance(v, (np.ndarray, torch.Tensor)):
                v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
                v = np.array2string(v, threshold=np.inf, max_line_width=np.inf, separator=",").replace("\n", "")
            else:
                v = np.round(v, NUM_DECIMALS_REPR)
            lines.append(f"{k}: {v}")
        return "\n".join(lines) + "\n"
