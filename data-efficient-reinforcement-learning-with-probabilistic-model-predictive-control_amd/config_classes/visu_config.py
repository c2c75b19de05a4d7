"""Same module path as the reference (rl_gp_mpc/config_classes/visu_config.py)."""
from .configs import VisuConfig  # noqa: F401
