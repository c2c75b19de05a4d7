"""Same module path as the reference (rl_gp_mpc/config_classes/reward_config.py)."""
from .configs import RewardConfig  # noqa: F401
