"""Assembly of one GP input row (same call surface as the reference's control_objects/utils/data_utils.py:4-9)."""
import torch


def form_model_input(state, action_model, time_idx, include_time_model, dim_input):
    """Row of the memory matrix X: the state, then the action, then -- for a time-varying model -- the control
    iteration as the last input dimension.  All fp64; the length must come out as ``dim_input``."""
    parts = [torch.as_tensor(state, dtype=torch.float64).reshape(-1),
             torch.as_tensor(action_model, dtype=torch.float64).reshape(-1)]
    if include_time_model:
        parts.append(torch.tensor([float(time_idx)], dtype=torch.float64))
    row = torch.cat(parts)
    if row.numel() != dim_input:
        raise ValueError(f"model input has {row.numel()} entries, the model expects {dim_input}")
    return row
