"""The transition-model plugin surface the controller programs against.

Mirrors what the reference's controller actually calls on its model (rl_gp_mpc/control_objects/models/
abstract_model.py:5-28 and the call sites gp_mpc_controller.py:117, 161, 202-227, 268), so that a model written for
either code base can be handed to either controller."""
import abc


class AbstractStateTransitionModel(abc.ABC):
    def __init__(self, config, dim_state, dim_action):
        self.config = config
        self.dim_state = int(dim_state)
        self.dim_action = int(dim_action)
        self.dim_input = self.dim_state + self.dim_action          # a time-varying model adds one input on top

    @abc.abstractmethod
    def prepare_inference(self, x, y):
        """Cache whatever prediction needs for the memory x (N, dim_input), y (N, dim_state)."""

    @abc.abstractmethod
    def predict_trajectory(self, actions, obs_mu, obs_var, len_horizon, current_time_idx):
        """actions (H, A), obs_mu (D,), obs_var (D, D) -> means (H + 1, D), covariances (H + 1, D, D); row 0 is the input."""

    @abc.abstractmethod
    def save_state(self):
        """Everything a training process needs (memory + hyper-parameters), picklable."""

    def load_state(self, saved_state):
        raise NotImplementedError(f"{type(self).__name__} does not restore saved states")

    def train(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} has no training routine")
