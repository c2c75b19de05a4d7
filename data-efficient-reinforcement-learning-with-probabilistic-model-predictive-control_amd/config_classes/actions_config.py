"""Same module path as the reference (rl_gp_mpc/config_classes/actions_config.py)."""
from .configs import ActionsConfig  # noqa: F401
