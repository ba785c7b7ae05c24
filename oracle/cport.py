"""ctypes wrapper of oracle/c/libbjxoracle.so (TEST INFRASTRUCTURE, see package docstring).

Used by tests (validated bit-for-bit against oracle/hmc.py) and by bench.py's
``cpu_baseline`` leg ("kind": "port").  Never imported by ``blackjax_amd``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "c", "libbjxoracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        f = lib.bjx_oracle_hmc_diag_gaussian
        f.restype = ctypes.c_int
        fp = ctypes.POINTER(ctypes.c_float)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        f.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int64, ctypes.c_int64,
                      ctypes.c_int64, ctypes.c_int, ctypes.c_float, fp, fp, ctypes.c_float,
                      fp, fp, fp, fp, u8, u8, ctypes.c_int]
        lib.bjx_oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def num_threads() -> int:
    return load().bjx_oracle_num_threads()


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def hmc_diag_gaussian_step(rng_key, q, logp, g, eps, imm, inv_var, L, thr=1000.0,
                           chain_offset=0, nthreads=0):
    """One transition in place on (q, logp, g); returns (acceptance_rate, is_accepted, is_divergent)."""
    lib = load()
    N, D = q.shape
    for a in (q, logp, g, imm, inv_var):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    acc = np.empty(N, np.float32)
    ia = np.empty(N, np.uint8)
    idv = np.empty(N, np.uint8)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    rc = lib.bjx_oracle_hmc_diag_gaussian(int(rng_key[0]), int(rng_key[1]), chain_offset, N, D,
                                          int(L), float(eps), _fp(imm), _fp(inv_var), float(thr),
                                          _fp(q), _fp(logp), _fp(g), _fp(acc),
                                          ia.ctypes.data_as(u8), idv.ctypes.data_as(u8), nthreads)
    assert rc == 0
    return acc, ia.astype(bool), idv.astype(bool)
