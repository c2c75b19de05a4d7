class BaseControllerObject:
    """Controller interface of the reference (controllers/abstract_controller.py)."""

    def add_memory(self, obs, action, obs_new, reward, **kwargs):
        raise NotImplementedError

    def get_action(self, obs_mu, obs_var=None):
        raise NotImplementedError
