"""Same module path as the reference (rl_gp_mpc/config_classes/observation_config.py)."""
from .configs import ObservationConfig  # noqa: F401
