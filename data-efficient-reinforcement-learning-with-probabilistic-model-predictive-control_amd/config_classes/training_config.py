"""Same module path as the reference (rl_gp_mpc/config_classes/training_config.py)."""
from .configs import TrainingConfig  # noqa: F401
