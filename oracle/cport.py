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
        f = lib.bjx_oracle_hmc_diag_gaussian_pc
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                      fp, fp, ctypes.c_int64, fp, ctypes.c_float, fp, fp, fp, fp, u8, u8, ctypes.c_int]
        f = lib.bjx_oracle_gemm_f32chain
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, fp, fp, ctypes.c_int64,
                      ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), fp, ctypes.c_int]
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


def hmc_diag_gaussian_step_pc(chain_keys, q, logp, g, eps, imm, inv_var, L, thr=1000.0, nthreads=0):
    """One transition in place on (q, logp, g) with explicit per-chain keys (N, 2), per-chain step
    sizes (N,) and a shared (D,) or per-chain (N, D) inverse mass diagonal; returns
    (acceptance_rate, is_accepted, is_divergent).  Bit-identical to oracle/hmc.py::kernel with
    ``chain_keys_override`` (tests/test_oracle_c.py)."""
    lib = load()
    N, D = q.shape
    keys = np.ascontiguousarray(chain_keys, dtype=np.uint32)
    assert keys.shape == (N, 2)
    eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, np.float32), (N,)))
    imm = np.ascontiguousarray(imm, dtype=np.float32)
    assert imm.shape in ((D,), (N, D))
    for a in (q, logp, g, inv_var):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    acc = np.empty(N, np.float32)
    ia = np.empty(N, np.uint8)
    idv = np.empty(N, np.uint8)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    rc = lib.bjx_oracle_hmc_diag_gaussian_pc(
        keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), N, D, int(L), _fp(eps), _fp(imm),
        D if imm.ndim == 2 else 0, _fp(inv_var), float(thr), _fp(q), _fp(logp), _fp(g), _fp(acc),
        ia.ctypes.data_as(u8), idv.ctypes.data_as(u8), nthreads)
    assert rc == 0
    return acc, ia.astype(bool), idv.astype(bool)


def gemm_f32chain(a, b_kn, k_order, nthreads=0):
    """C[m][n] = fp32 fmaf chain over k in ``k_order`` of a[m][k] * b_kn[k][n] (``b_kn`` may be any
    strided 2-d view, e.g. ``mat.T``)."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    M, K = a.shape
    assert b_kn.dtype == np.float32 and b_kn.shape[0] == K
    Nn = b_kn.shape[1]
    sbk, sbn = b_kn.strides[0] // 4, b_kn.strides[1] // 4
    order = np.ascontiguousarray(k_order, dtype=np.int32)
    assert sorted(order.tolist()) == list(range(K))
    c = np.empty((M, Nn), np.float32)
    ptr = ctypes.cast(b_kn.ctypes.data, ctypes.POINTER(ctypes.c_float))
    rc = lib.bjx_oracle_gemm_f32chain(M, K, Nn, _fp(a), ptr, sbk, sbn,
                                      order.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(c),
                                      nthreads)
    assert rc == 0
    return c
