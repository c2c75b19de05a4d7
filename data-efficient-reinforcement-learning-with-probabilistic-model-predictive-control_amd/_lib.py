"""ctypes binding of libgpmpc_hip.so (C ABI declared in include/gpmpc.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing
this module raises.  Build it with ``python __graft_entry__.py`` (or ``make -C <pkg>/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPMPC_LIB") or os.path.join(_HERE, "libgpmpc_hip.so")   # GPMPC_LIB: debug builds

GPMPC_OK = 0
GPMPC_ERR_ARG = -1
GPMPC_ERR_NOT_PD = -2
GPMPC_ERR_HIP = -3
GPMPC_ERR_LIMIT = -4


class GpmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpmpc error {code}: {msg}")
        self.code = code


class NotPositiveDefiniteError(GpmpcError):
    """Cholesky failure of K + noise*I -- the reference lets torch.linalg.cholesky's error
    propagate uncaught (rl_gp_mpc/control_objects/models/gp_model.py:427)."""


_P = C.c_void_p
_D = C.c_double
_I = C.c_int

SIGNATURES = {
    "gpmpc_abi_version": (C.c_int, []),
    "gpmpc_create": (C.c_int, [C.POINTER(_P), _I]),
    "gpmpc_destroy": (C.c_int, [_P]),
    "gpmpc_last_error": (C.c_char_p, [_P]),
    "gpmpc_prepare": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gpmpc_set_factors": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gpmpc_get_factors": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "gpmpc_read_factors": (C.c_int, [_P, _P, _P, _P]),
    "gpmpc_last_prepare_mode": (C.c_int, [_P]),
    "gpmpc_last_rollout_path": (C.c_int, [_P]),
    "gpmpc_last_grad_path": (C.c_int, [_P]),
    "gpmpc_last_cluster": (C.c_int, [_P]),
    "gpmpc_build_id": (C.c_char_p, []),
    "gpmpc_mll": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(_D), _P]),
    "gpmpc_set_option": (C.c_int, [_P, C.c_char_p, C.c_longlong]),
    "gpmpc_set_cost": (C.c_int, [_P, _P, _P, _P, _D, _I, _P, _P, _I, _I]),
    "gpmpc_rollout": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P]),
    "gpmpc_rollout_grad": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P]),
    "gpmpc_objective_grad_host": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _D, C.POINTER(C.POINTER(_D)), _P]),
    "gpmpc_rollout_timed": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _I, C.POINTER(C.c_float), _P]),
    "gpmpc_cem_search": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _D, _I, _I, C.c_ulonglong, _P, _I, _P, _P, _P, _P, _P]),
    "gpmpc_cem_local": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _D, _I, _I, C.c_ulonglong, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gpmpc_cem_merge": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "gpmpc_argmin_async": (C.c_int, [_P, _P, _I, C.c_longlong, _P, _I, _P, _P]),
    "gpmpc_argmin": (C.c_int, [_P, _P, _I, C.c_longlong, C.POINTER(_D), C.POINTER(C.c_longlong), _P]),
}


ABI_VERSION = 12


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python __graft_entry__.py` or `make -C <package>/csrc`.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gpmpc_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.gpmpc_abi_version()} != expected {ABI_VERSION}; rebuild the HIP library")
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
