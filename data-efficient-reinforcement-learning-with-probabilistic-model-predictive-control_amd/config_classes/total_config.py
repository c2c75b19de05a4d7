"""Same module path as the reference (rl_gp_mpc/config_classes/total_config.py)."""
from .configs import Config  # noqa: F401
