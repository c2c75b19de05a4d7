"""MI355X-native GP-MPC inner loop (drop-in for the hot path of
SimonRennotte/Data-Efficient-Reinforcement-Learning-with-Probabilistic-Model-Predictive-Control).

Arithmetic lives in hand-written HIP kernels for gfx950 behind the C ABI of include/gpmpc.h
(csrc/ -> libgpmpc_hip.so); this package is the host-side mirror of the reference's
transition-model / controller interface for that path.
"""
from ._lib import GpmpcError, NotPositiveDefiniteError, LIB_PATH  # noqa: F401
from .engine import HipEngine  # noqa: F401
from .control_objects.controllers.gp_mpc_controller import GpMpcController  # noqa: F401,E402
from .control_objects.models.gp_model import GpStateTransitionModel  # noqa: F401,E402
