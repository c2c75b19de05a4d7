"""Same module path as the reference (rl_gp_mpc/config_classes/controller_config.py)."""
from .configs import ControllerConfig  # noqa: F401
