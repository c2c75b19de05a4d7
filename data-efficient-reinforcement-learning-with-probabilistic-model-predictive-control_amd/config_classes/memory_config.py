"""Same module path as the reference (rl_gp_mpc/config_classes/memory_config.py)."""
from .configs import MemoryConfig  # noqa: F401
