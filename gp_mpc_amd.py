"""Importable alias for the package directory (its mandated name contains hyphens):
``import gp_mpc_amd`` == the package
``data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd``."""
import importlib
import os
import sys

_PKG = "data-efficient-reinforcement-learning-with-probabilistic-model-predictive-control_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_mod = importlib.import_module(_PKG)
sys.modules[__name__] = _mod
sys.modules.setdefault("gp_mpc_amd", _mod)
