import torch


def form_model_input(state, action_model, time_idx, include_time_model, dim_input):
    """GP input row [state, action, (time)]  (reference control_objects/utils/data_utils.py:4-9)."""
    x = torch.empty(dim_input, dtype=torch.float64)
    n = state.shape[0] + action_model.shape[0]
    x[:n] = torch.cat((state, action_model))
    if include_time_model:
        x[-1] = time_idx
    return x
