"""Restatement of JAX's threefry PRNG (TEST INFRASTRUCTURE, see package docstring).

Third-party arithmetic that is NOT under /root/reference: module ``jax`` pinned
``0.10.0`` (uv.lock:1309-1311; floor ``jax>=0.9.0`` pyproject.toml:34-35).
Hot-path call sites in the reference: ``blackjax/mcmc/hmc.py:299`` and
``nuts.py:133`` (``split(key, 2)``), ``util.py:90`` (``normal``),
``proposal.py:123,156,226`` (``bernoulli``), ``trajectory.py:321,645``
(``fold_in``), ``trajectory.py:646`` (``split(., 3)``), ``trajectory.py:650``
(``bernoulli(key)``), ``util.py:203`` and
``adaptation/staged_adaptation.py:868`` (``split(key, T)``).

Published algorithm restated (jax/_src/prng.py, jax/_src/random.py, with the
``jax_threefry_partitionable=True`` default of jax >= 0.5):

* key          ``uint32[2]``; ``key(seed) = [seed >> 32, seed & 0xffffffff]``
* threefry2x32 Salmon et al. 2011 (Random123), 20 rounds, rotations
               ``[13,15,26,6] / [17,29,16,24]``, key schedule injected every 4 rounds
* split(k, n)[i] = fold_in(k, i) = threefry(k, (hi32(i), lo32(i)))  (both words)
* random_bits(k, 32, shape)[i] = o0 ^ o1 of threefry(k, (hi32(i), lo32(i))),
               i = row-major flat index
* uniform f32  ``f = bitcast((bits >> 9) | 0x3f800000) - 1``;
               ``u = max(minval, f*(maxval-minval) + minval)``
* bernoulli    ``uniform(key, shape) < p``
* normal f32   ``u = uniform(key, shape, minval=nextafter(-1, 0), maxval=1)``;
               ``z = sqrt(2) * erf_inv(u)``; XLA's f32 ``erf_inv`` is Giles'
               single-precision polynomial (xla/client/lib/math.cc ErfInv32).

The block function is pinned by the Random123 known-answer vectors
(tests/test_oracle_prng.py).  The derived streams are "parity unpinned" (no
literal jax.random outputs exist in the reference's tests and JAX cannot be
imported here).
"""
from __future__ import annotations

import numpy as np

from .fp import f32, fma32, log1p_cr, sqrt32

u32 = np.uint32

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = u32(0x1BD11BDA)


def _rotl(x, r):
    return (x << u32(r)) | (x >> u32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds.  All arguments uint32 arrays (broadcast)."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(k0, dtype=u32)
        k1 = np.asarray(k1, dtype=u32)
        x0 = np.asarray(x0, dtype=u32).copy()
        x1 = np.asarray(x1, dtype=u32).copy()
        ks = (k0, k1, k0 ^ k1 ^ _PARITY)
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + u32(i + 1)
        return x0, x1


def key(seed: int) -> np.ndarray:
    """jax.random.key / PRNGKey(seed) -> uint32[2] (threefry_seed)."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=u32)


def as_key(k) -> np.ndarray:
    if isinstance(k, (int, np.integer)):
        return key(int(k))
    k = np.asarray(k)
    assert k.shape[-1] == 2
    return k.astype(u32)


def _counts(n: int, offset: int = 0):
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    return (idx >> np.uint64(32)).astype(u32), (idx & np.uint64(0xFFFFFFFF)).astype(u32)


def split(k, n: int = 2, offset: int = 0) -> np.ndarray:
    """jax.random.split(key, n) -> (n, 2).  ``k`` may be a batch (..., 2) of keys,
    the result is then (..., n, 2).  ``offset`` selects rows offset..offset+n of a
    larger split (used for chain sharding)."""
    k = as_key(k)
    hi, lo = _counts(n, offset)
    o0, o1 = threefry2x32(k[..., 0:1], k[..., 1:2], hi, lo)
    return np.stack([o0, o1], axis=-1)


def split_at(k, indices) -> np.ndarray:
    """Rows ``indices`` of ``jax.random.split(k, n)`` for any n > max(indices): child keys of ONE
    key at arbitrary (64-bit) positions -- the per-chain keys of a subset of global chain indices."""
    k = as_key(k)
    idx = np.asarray(indices, dtype=np.uint64)
    hi = (idx >> np.uint64(32)).astype(u32)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(u32)
    o0, o1 = threefry2x32(k[0], k[1], hi, lo)
    return np.stack([o0, o1], axis=-1)


def fold_in(k, data) -> np.ndarray:
    """jax.random.fold_in(key, data): threefry(key, (0, data)); batched over keys/data."""
    k = as_key(k)
    data = np.asarray(data, dtype=u32)
    o0, o1 = threefry2x32(k[..., 0], k[..., 1], u32(0), data)
    return np.stack([o0, o1], axis=-1)


def random_bits(k, shape) -> np.ndarray:
    """32 random bits per element.  ``k`` is one key (2,) or a batch (B, 2); the
    result has shape ``shape`` or ``(B,) + shape``."""
    k = as_key(k)
    shape = tuple(shape)
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    hi, lo = _counts(n)
    o0, o1 = threefry2x32(k[..., 0:1], k[..., 1:2], hi, lo)
    bits = o0 ^ o1
    return bits.reshape(k.shape[:-1] + shape)


def bits_to_unit_float(bits) -> np.ndarray:
    """[0, 1) float from the top 23 bits (jax/_src/random.py::_uniform)."""
    fb = (np.asarray(bits, dtype=u32) >> u32(9)) | u32(0x3F800000)
    return fb.view(f32) - f32(1.0)


def uniform(k, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    f = bits_to_unit_float(random_bits(k, shape))
    minval = f32(minval)
    maxval = f32(maxval)
    scale = f32(maxval - minval)
    return np.maximum(minval, fma32(f, scale, minval))


def bernoulli(k, p=0.5, shape=None) -> np.ndarray:
    p = np.asarray(p, dtype=f32)
    if shape is None:
        shape = ()
    return uniform(k, shape) < p


# xla/client/lib/math.cc::ErfInv32 (Giles, "Approximating the erfinv function")
_ERFINV_LT5 = np.array(
    [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
     -0.00125372503, -0.00417768164, 0.246640727, 1.50140941], dtype=f32)
_ERFINV_GE5 = np.array(
    [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
     -0.0076224613, 0.00943887047, 1.00167406, 2.83297682], dtype=f32)


def erf_inv(x) -> np.ndarray:
    x = np.asarray(x, dtype=f32)
    w = -log1p_cr(-(x * x))
    lt = w < f32(5.0)
    with np.errstate(invalid="ignore"):
        w = np.where(lt, w - f32(2.5), sqrt32(w) - f32(3.0)).astype(f32)
    p = np.where(lt, _ERFINV_LT5[0], _ERFINV_GE5[0]).astype(f32)
    for i in range(1, 9):
        c = np.where(lt, _ERFINV_LT5[i], _ERFINV_GE5[i]).astype(f32)
        p = fma32(p, w, c)
    res = p * x
    return np.where(np.abs(x) == f32(1.0), x * f32(np.inf), res).astype(f32)


_NORMAL_LO = np.nextafter(f32(-1.0), f32(0.0), dtype=f32)
_SQRT2 = f32(np.sqrt(2))


def normal(k, shape=()) -> np.ndarray:
    """jax.random.normal(key, shape, float32)."""
    u = uniform(k, shape, minval=_NORMAL_LO, maxval=f32(1.0))
    return (_SQRT2 * erf_inv(u)).astype(f32)


def randint(k, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint(key, (), minval, maxval, int32) for one key or a batch (..., 2) of keys.
    Restated from jax/_src/random.py::_randint: two 32-bit draws from split(key, 2); with
    span = maxval - minval (1 if maxval <= minval) and multiplier = 2^32 % span (computed as
    ((2^16 % span)^2) % span): offset = ((hi % span) * multiplier + lo % span) % span."""
    k = as_key(k)
    kk = split(k, 2)
    hi = random_bits(kk[..., 0, :], ())
    lo = random_bits(kk[..., 1, :], ())
    span = np.uint32(maxval - minval) if maxval > minval else np.uint32(1)
    mult = np.uint32(1 << 16) % span
    mult = (mult * mult) % span
    with np.errstate(over="ignore"):
        off = ((hi % span) * mult + (lo % span)) % span
    return (np.int32(minval) + off.astype(np.int32)).astype(np.int32)


def permutation(k, n: int) -> np.ndarray:
    """jax.random.permutation(key, n) for an integer n (jax/_src/random.py::_shuffle): repeated
    stable sorts by fresh 32-bit keys; rounds = ceil(3 ln n / ln(2^32 - 1)).  Each round
    ``key, subkey = split(key)``; ``sort_keys = random_bits(subkey, 32, (n,))``."""
    k = as_key(k)
    x = np.arange(n, dtype=np.int64)
    rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(float(2**32 - 1))))
    for _ in range(rounds):
        kk = split(k, 2)
        k, sub = kk[0], kk[1]
        bits = random_bits(sub, (n,))
        x = x[np.argsort(bits, kind="stable")]
    return x
