"""Same module path as the reference (rl_gp_mpc/config_classes/model_config.py)."""
from .configs import ModelConfig  # noqa: F401
