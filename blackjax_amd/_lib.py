"""ctypes binding of ``libbjxhip.so`` (C ABI declared in ``include/bjx_hip.h``).

There is NO fallback: if the shared library is missing or a call fails the
product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C blackjax_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbjxhip.so")

_f32p = c_void_p  # device pointers travel as integers (tensor.data_ptr())
_u8p = c_void_p

# name -> argtypes ; every function returns int except bjx_last_error
SIGNATURES = {
    "bjx_abi_version": [],
    "bjx_keys_split": [c_uint32, c_uint32, c_int64, c_int64, POINTER(c_uint32)],
    "bjx_rng_normal": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, _f32p],
    "bjx_rng_uniform": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, _f32p],
    "bjx_hmc_momentum_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, _f32p,
                              c_int64, _f32p, _f32p],
    "bjx_leapfrog_diag": [c_void_p, c_int64, c_int64, c_int, c_float, _f32p, _f32p, c_int64,
                          _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_hmc_finish_diag": [c_void_p, c_uint32, c_uint32, c_int64, c_int64, c_int64, c_int64, c_float,
                            _f32p, _f32p, c_int64, c_float,
                            _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                            _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _u8p, _f32p],
    "bjx_da_init": [c_void_p, c_int64, c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_da_update": [c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, _f32p, _f32p,
                      _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_exp": [c_void_p, c_int64, _f32p, _f32p],
    "bjx_welford_update_diag": [c_void_p, c_int64, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p, _f32p],
    "bjx_welford_final_diag": [c_void_p, c_int64, c_int64, c_int64, c_float, _f32p, _f32p, c_int64,
                               _f32p],
    "bjx_target_diag_gaussian": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p, _f32p],
    "bjx_target_neal_funnel": [c_void_p, c_int64, c_int64, _f32p, _f32p, _f32p],
    "bjx_target_ar1_gaussian": [c_void_p, c_int64, c_int64, c_float, c_float, c_float, _f32p,
                                _f32p, _f32p],
}

_lib = None


class BjxError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libbjxhip.so (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BjxError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C blackjax_amd/csrc`). "
            "blackjax_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.bjx_last_error.restype = c_char_p
    lib.bjx_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


class LaunchTimer:
    """Brackets selected entry points with HIP events on the current stream so a caller
    (bench.py) can read per-launch kernel durations of the timed region afterwards."""

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in self.names}

    def durations_ms(self, name):
        import torch

        torch.cuda.synchronize()
        return [s.elapsed_time(e) for s, e in self.events[name]]


_timer = None


def set_timer(timer) -> None:
    global _timer
    _timer = timer


def call(name: str, *args) -> None:
    lib = load()
    if _timer is not None and name in _timer.names:
        import torch

        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        _timer.events[name].append((s, e))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise BjxError(f"{name} failed (rc={rc}): {lib.bjx_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
