"""``jax.random`` stand-in: the threefry restatement of ``oracle/prng.py`` (jax 0.10 defaults, partitionable layout).
Keys are plain ``uint32[..., 2]`` NumPy arrays.  THIS IS NOT JAX'S RNG: the bit streams are the oracle's, so nothing
generated through this module pins SURVEY row a34 -- only the way the reference CONSUMES keys is exercised."""
from __future__ import annotations

import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from oracle import prng as _p  # noqa: E402

from . import _Missing, asarray  # noqa: E402


def __getattr__(item):
    if item.startswith("__") and item.endswith("__"):
        raise AttributeError(item)
    return _Missing(f"jax.random.{item}")


def _key(k):
    return _p.as_key(np.asarray(k))


def key(seed):
    return _p.key(int(seed))


PRNGKey = key


def key_data(k):
    return _key(k)


def wrap_key_data(data, impl=None):
    return _key(data)


def split(k, num=2):
    return _p.split(_key(k), int(num))


def fold_in(k, data):
    d = data.item() if hasattr(data, "item") else data
    return _p.fold_in(_key(k), np.uint32(int(d) & 0xFFFFFFFF))


def _shape(shape):
    if shape is None:
        return ()
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


def bits(k, shape=(), dtype=None):
    return asarray(_p.random_bits(_key(k), _shape(shape)).astype(np.int64))


def uniform(k, shape=(), dtype=None, minval=0.0, maxval=1.0):
    lo = float(minval.item() if hasattr(minval, "item") else minval)
    hi = float(maxval.item() if hasattr(maxval, "item") else maxval)
    return asarray(_p.uniform(_key(k), _shape(shape), lo, hi))


def normal(k, shape=(), dtype=None):
    return asarray(_p.normal(_key(k), _shape(shape)))


def bernoulli(k, p=0.5, shape=None):
    pa = np.asarray(p.detach().numpy() if hasattr(p, "detach") else p, dtype=np.float32)
    shp = pa.shape if shape is None else _shape(shape)
    return asarray(np.asarray(_p.uniform(_key(k), shp) < pa))


def randint(k, shape, minval, maxval, dtype=None):
    if _shape(shape) != ():
        raise NotImplementedError("refshim randint: scalar draws only")
    return asarray(np.int32(_p.randint(_key(k), int(minval), int(maxval))))


def permutation(k, x, axis=0, independent=False):
    if isinstance(x, (int, np.integer)):
        return asarray(_p.permutation(_key(k), int(x)).astype(np.int32))
    idx = _p.permutation(_key(k), int(x.shape[0]))
    return x[idx.astype(np.int64)]


def choice(k, a, shape=(), replace=True, p=None, axis=0):
    """jax/_src/random.py::choice for the scalar / with-replacement uses of the reference's tests: without ``p`` an index
    from ``randint(key, shape, 0, n)``; with ``p`` the inverse-CDF draw ``searchsorted(cumsum(p), p_total * (1 - uniform))``."""
    arr = asarray(np.arange(int(a))) if isinstance(a, (int, np.integer)) else asarray(a)
    n = int(arr.shape[0])
    shp = _shape(shape)
    if not replace:
        raise NotImplementedError("refshim choice: replace=False")
    if p is None:
        if shp != ():
            raise NotImplementedError("refshim choice: scalar draws only without p")
        return arr[int(_p.randint(_key(k), 0, n))]
    pc = np.cumsum(np.asarray(p.detach().numpy() if hasattr(p, "detach") else p, dtype=np.float32), dtype=np.float32)
    r = pc[-1] * (np.float32(1.0) - _p.uniform(_key(k), shp))
    ind = np.searchsorted(pc, r, side="left")
    return arr[asarray(np.asarray(ind, dtype=np.int64))] if shp != () else arr[int(ind)]
