"""Host-side threefry key helpers with ``jax.random`` semantics.

Keys are ``uint32[2]`` (jax.random key data).  Device kernels derive per-chain and
per-element streams themselves from a key passed by value; the host only ever
splits a run key into per-step keys (blackjax/util.py:203,
adaptation/staged_adaptation.py:868), which is done by the C-ABI host function
``bjx_keys_split``.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

__all__ = ["key", "PRNGKey", "split", "fold_in", "uniform", "key_words", "ChainMajorKey", "key_spec"]


def key(seed: int) -> np.ndarray:
    """jax.random.key(seed) key data: ``[seed >> 32, seed & 0xffffffff]``."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


PRNGKey = key


def key_words(rng_key) -> tuple[int, int]:
    """Normalise an int seed / array-like uint32[2] key to two Python ints."""
    if isinstance(rng_key, (int, np.integer)):
        rng_key = key(int(rng_key))
    if hasattr(rng_key, "detach"):  # torch tensor (must be host data: no device sync here)
        if rng_key.is_cuda:
            raise ValueError("rng_key must live on the host (uint32[2]); got a device tensor")
        rng_key = rng_key.detach().numpy()
    k = np.asarray(rng_key)
    if k.shape != (2,):
        raise ValueError(f"rng_key must be an int seed or have shape (2,), got {k.shape}")
    return int(k[0]) & 0xFFFFFFFF, int(k[1]) & 0xFFFFFFFF


def split(rng_key, num: int = 2, offset: int = 0) -> np.ndarray:
    """jax.random.split(key, num) -> (num, 2) uint32 (rows offset..offset+num of a larger split)."""
    k0, k1 = key_words(rng_key)
    out = np.empty((int(num), 2), dtype=np.uint32)
    _lib.call("bjx_keys_split", k0, k1, int(num), int(offset),
              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return out


def fold_in(rng_key, data: int) -> np.ndarray:
    """jax.random.fold_in(key, data) == split(key, .)[data] for threefry-partitionable keys."""
    return split(rng_key, 1, offset=int(data) & 0xFFFFFFFF)[0]


def uniform(rng_key) -> np.float32:
    """jax.random.uniform(key, (), float32) on the host: 23 mantissa bits of
    ``random_bits(key, 32, ())`` = the xor of the two threefry output words at counter 0."""
    w = split(rng_key, 1)[0]
    bits = np.uint32(w[0] ^ w[1])
    f = (np.array([(bits >> np.uint32(9)) | np.uint32(0x3F800000)], np.uint32).view(np.float32)[0]
         - np.float32(1.0))
    return np.float32(max(np.float32(0.0), f))


class ChainMajorKey:
    """Step ``t`` of a chain-major run: chain ``i`` uses
    ``split(split(run_key, N)[i], T)[t]`` -- the key layout of a vmapped per-chain loop such
    as ``jax.vmap(window_adaptation(...).run)(jax.random.split(key, N), positions)``.
    Accepted wherever a kernel takes ``rng_key``."""

    __slots__ = ("run_key", "step")

    def __init__(self, run_key, step: int):
        self.run_key = run_key
        self.step = int(step)


def key_spec(rng_key) -> tuple[int, int, int]:
    """-> (key word 0, key word 1, step_fold) with step_fold = -1 for a plain (step-major) key."""
    if isinstance(rng_key, ChainMajorKey):
        k0, k1 = key_words(rng_key.run_key)
        return k0, k1, rng_key.step
    k0, k1 = key_words(rng_key)
    return k0, k1, -1
