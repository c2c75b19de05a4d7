class AbstractStateTransitionModel:
    """The reference's plugin surface for a transition model
    (rl_gp_mpc/control_objects/models/abstract_model.py:5-28)."""

    def __init__(self, config, dim_state, dim_action):
        self.config = config
        self.dim_state = dim_state
        self.dim_action = dim_action
        self.dim_input = dim_state + dim_action

    def predict_trajectory(self, actions, obs_mu, obs_var, len_horizon, current_time_idx):
        raise NotImplementedError

    def prepare_inference(self, x, y):
        raise NotImplementedError

    def train(self, *args, **kwargs):
        raise NotImplementedError

    def save_state(self):
        raise NotImplementedError

    def load_state(self, saved_state):
        raise NotImplementedError
