"""``jax.numpy`` stand-in on torch CPU tensors (see ``tests/refshim/jax/__init__.py``: NOT JAX)."""
from __future__ import annotations

import builtins
import math
import sys
import types

import numpy as _np
import torch

from .. import Array, _Missing, _np_dtype_to_torch, _wrap, asarray

float32, float64, int32, int64, uint32, bool_ = torch.float32, torch.float64, torch.int32, torch.int64, torch.int64, torch.bool
int8, uint8, int16, float16, bfloat16 = torch.int8, torch.uint8, torch.int16, torch.float16, torch.bfloat16
inf, pi, nan, e, newaxis = math.inf, math.pi, math.nan, math.e, None
ndarray = Array
array = asarray
integer, floating, number = _np.integer, _np.floating, _np.number


def __getattr__(item):
    if item.startswith("__") and item.endswith("__"):
        raise AttributeError(item)
    return _Missing(f"jax.numpy.{item}")


def _t(x):
    return asarray(x).as_subclass(torch.Tensor)


def _shape(shape):
    if isinstance(shape, (int, _np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


def zeros(shape, dtype=None):
    return _wrap(torch.zeros(_shape(shape), dtype=_np_dtype_to_torch(dtype) or torch.float32))


def ones(shape, dtype=None):
    return _wrap(torch.ones(_shape(shape), dtype=_np_dtype_to_torch(dtype) or torch.float32))


def full(shape, fill_value, dtype=None):
    fv = asarray(fill_value)
    return _wrap(torch.full(_shape(shape), fv.item(), dtype=_np_dtype_to_torch(dtype) or fv.dtype))


def zeros_like(x, dtype=None):
    return _wrap(torch.zeros_like(_t(x), dtype=_np_dtype_to_torch(dtype)))


def ones_like(x, dtype=None):
    return _wrap(torch.ones_like(_t(x), dtype=_np_dtype_to_torch(dtype)))


def full_like(x, fill_value, dtype=None):
    return _wrap(torch.full_like(_t(x), asarray(fill_value).item(), dtype=_np_dtype_to_torch(dtype)))


def empty(shape, dtype=None):
    return zeros(shape, dtype)


def arange(start, stop=None, step=1, dtype=None):
    args = [int(start)] if stop is None else [int(start), int(stop), int(step)]
    return _wrap(torch.arange(*args, dtype=_np_dtype_to_torch(dtype) or torch.int32))


def linspace(start, stop, num=50, dtype=None):
    return _wrap(torch.linspace(float(start), float(stop), int(num), dtype=_np_dtype_to_torch(dtype) or torch.float32))


def eye(n, dtype=None):
    return _wrap(torch.eye(int(n), dtype=_np_dtype_to_torch(dtype) or torch.float32))


identity = eye


def _unary(fn):
    def f(x):
        t = _t(x)
        if not t.is_floating_point() and fn not in (torch.abs, torch.neg, torch.sign, torch.logical_not):
            t = t.to(torch.float32)
        return _wrap(fn(t))

    return f


exp, log, sqrt, log1p, expm1 = _unary(torch.exp), _unary(torch.log), _unary(torch.sqrt), _unary(torch.log1p), _unary(torch.expm1)
sin, cos, tanh, floor, ceil = _unary(torch.sin), _unary(torch.cos), _unary(torch.tanh), _unary(torch.floor), _unary(torch.ceil)
abs = absolute = _unary(torch.abs)
negative, sign, square = _unary(torch.neg), _unary(torch.sign), _unary(torch.square)
isfinite = lambda x: _wrap(torch.isfinite(_t(x)))  # noqa: E731
isnan = lambda x: _wrap(torch.isnan(_t(x)))  # noqa: E731
isinf = lambda x: _wrap(torch.isinf(_t(x)))  # noqa: E731
logical_not = lambda x: _wrap(torch.logical_not(_t(x)))  # noqa: E731
rint = round = lambda x: _wrap(torch.round(_t(x)))  # noqa: E731  (half to even, like jnp.rint)


def _binary(fn):
    def f(x, y):
        return _wrap(fn(_t(x), _t(y)))

    return f


add, subtract, multiply, divide = _binary(torch.add), _binary(torch.sub), _binary(torch.mul), _binary(torch.true_divide)
true_divide = divide
minimum, maximum, power = _binary(torch.minimum), _binary(torch.maximum), _binary(torch.pow)
logaddexp = _binary(torch.logaddexp)
logical_and, logical_or = _binary(torch.logical_and), _binary(torch.logical_or)
equal, not_equal, less, greater = _binary(torch.eq), _binary(torch.ne), _binary(torch.lt), _binary(torch.gt)
less_equal, greater_equal = _binary(torch.le), _binary(torch.ge)
mod = remainder = _binary(torch.remainder)
floor_divide = lambda x, y: _wrap(torch.div(_t(x), _t(y), rounding_mode="floor"))  # noqa: E731
bitwise_and, bitwise_or, bitwise_xor = _binary(torch.bitwise_and), _binary(torch.bitwise_or), _binary(torch.bitwise_xor)
right_shift, left_shift = _binary(torch.bitwise_right_shift), _binary(torch.bitwise_left_shift)


def bitwise_count(x):
    t = _t(x).to(torch.int64)
    out = torch.zeros_like(t)
    for b in range(63):
        out += (t >> b) & 1
    return _wrap(out.to(_t(x).dtype))


def where(cond, x=None, y=None):
    if x is None:
        return tuple(_wrap(v) for v in torch.where(_t(cond)))
    xt, yt = _t(x), _t(y)
    if xt.dtype != yt.dtype:
        dt = torch.promote_types(xt.dtype, yt.dtype)
        # Python scalars are weak in JAX: a float32 array and the scalar -inf stay float32
        if isinstance(x, (int, float)) and not isinstance(y, (int, float)):
            dt = yt.dtype if (yt.is_floating_point() or not isinstance(x, float)) else torch.float32
        if isinstance(y, (int, float)) and not isinstance(x, (int, float)):
            dt = xt.dtype if (xt.is_floating_point() or not isinstance(y, float)) else torch.float32
        xt, yt = xt.to(dt), yt.to(dt)
    return _wrap(torch.where(_t(cond).to(torch.bool), xt, yt))


def clip(x, min=None, max=None, a_min=None, a_max=None):
    lo = min if min is not None else a_min
    hi = max if max is not None else a_max
    t = _t(x)
    return _wrap(torch.clamp(t, None if lo is None else _t(lo).to(t.dtype), None if hi is None else _t(hi).to(t.dtype)))


def _axis(axis):
    return None if axis is None else (tuple(axis) if isinstance(axis, (tuple, list)) else int(axis))


def sum(x, axis=None, keepdims=False, dtype=None, where=None):
    t = _t(x)
    if t.dtype == torch.bool:
        t = t.to(torch.int32)
    if where is not None:  # masked reduction: the excluded elements contribute zero
        t = torch.where(_t(where).to(torch.bool), t, torch.zeros((), dtype=t.dtype))
    return _wrap(torch.sum(t) if axis is None and not keepdims else torch.sum(t, dim=_axis(axis), keepdim=keepdims))


def mean(x, axis=None, keepdims=False, where=None):
    t = _t(x)
    t = t if t.is_floating_point() else t.to(torch.float32)
    if where is not None:  # sum over the selected elements / their count (nan when none is selected, as in NumPy / JAX)
        m = torch.broadcast_to(_t(where).to(torch.bool), t.shape)
        num = sum(t, axis, keepdims, where=m)
        den = sum(m.to(t.dtype), axis, keepdims)
        return _wrap(_t(num) / _t(den))
    return _wrap(torch.mean(t) if axis is None and not keepdims else torch.mean(t, dim=_axis(axis), keepdim=keepdims))


def var(x, axis=None, ddof=0, keepdims=False):
    t = _t(x)
    return _wrap(torch.var(t, correction=ddof) if axis is None else torch.var(t, dim=_axis(axis), correction=ddof, keepdim=keepdims))


def std(x, axis=None, ddof=0, keepdims=False):
    return sqrt(var(x, axis, ddof, keepdims))


def prod(x, axis=None):
    return _wrap(torch.prod(_t(x)) if axis is None else torch.prod(_t(x), dim=int(axis)))


def max(x, axis=None, keepdims=False):
    return _wrap(torch.max(_t(x)) if axis is None else torch.amax(_t(x), dim=_axis(axis), keepdim=keepdims))


def min(x, axis=None, keepdims=False):
    return _wrap(torch.min(_t(x)) if axis is None else torch.amin(_t(x), dim=_axis(axis), keepdim=keepdims))


amax, amin = max, min


def argmax(x, axis=None):
    return _wrap(torch.argmax(_t(x)) if axis is None else torch.argmax(_t(x), dim=int(axis)))


def argmin(x, axis=None):
    return _wrap(torch.argmin(_t(x)) if axis is None else torch.argmin(_t(x), dim=int(axis)))


def any(x, axis=None, keepdims=False):
    t = _t(x).to(torch.bool)
    return _wrap(torch.any(t) if axis is None and not keepdims else
                 torch.any(t, dim=tuple(range(t.ndim)) if axis is None else _axis(axis), keepdim=keepdims))


def all(x, axis=None, keepdims=False):
    t = _t(x).to(torch.bool)
    return _wrap(torch.all(t) if axis is None and not keepdims else
                 torch.all(t, dim=tuple(range(t.ndim)) if axis is None else _axis(axis), keepdim=keepdims))


def repeat(x, repeats, axis=None, total_repeat_length=None):
    t = _t(x)
    r = repeats if isinstance(repeats, int) else _t(repeats).long()
    return _wrap(torch.repeat_interleave(t.reshape(-1) if axis is None else t, r, dim=None if axis is None else int(axis)))


def average(x, axis=None, weights=None):
    if weights is None:
        return mean(x, axis)
    t, w = _t(x), _t(weights)
    if axis is None:
        return _wrap(torch.sum(t * w) / torch.sum(w))
    shape = [1] * t.ndim
    shape[int(axis)] = -1
    wv = w.reshape(shape) if w.ndim == 1 else w
    return _wrap(torch.sum(t * wv, dim=int(axis)) / torch.sum(wv, dim=int(axis)))


def nanmean(x, axis=None):
    return _wrap(torch.nanmean(_t(x)) if axis is None else torch.nanmean(_t(x), dim=_axis(axis)))


def median(x, axis=None):
    t = _t(x)
    return quantile(t, 0.5, axis=axis)  # numpy semantics (mean of the two middle values), unlike torch.median


def quantile(x, q, axis=None, method="linear"):
    t = _t(x)
    t = t if t.is_floating_point() else t.to(torch.float32)
    qt = _t(q).to(t.dtype)
    return _wrap(torch.quantile(t.reshape(-1) if axis is None else t, qt, dim=None if axis is None else int(axis), interpolation=method))


def moveaxis(x, source, destination):
    return _wrap(torch.movedim(_t(x), source, destination))


def indices(dimensions, dtype=None):
    grids = torch.meshgrid(*[torch.arange(int(n), dtype=_np_dtype_to_torch(dtype) or torch.int32) for n in dimensions], indexing="ij")
    return _wrap(torch.stack(grids))


def conjugate(x):
    return _wrap(torch.conj(_t(x)).resolve_conj())


conj = conjugate
real = lambda x: _wrap(torch.real(_t(x)))  # noqa: E731
imag = lambda x: _wrap(torch.imag(_t(x)))  # noqa: E731


def cumsum(x, axis=0):
    return _wrap(torch.cumsum(_t(x), dim=int(axis)))


def dot(a, b, precision=None, preferred_element_type=None):
    at, bt = _t(a), _t(b)
    if at.ndim == 0 or bt.ndim == 0:
        return _wrap(at * bt)
    if at.ndim == 1 and bt.ndim == 1:
        return _wrap(torch.dot(at, bt))
    return _wrap(torch.matmul(at, bt))


def matmul(a, b, precision=None):
    return _wrap(torch.matmul(_t(a), _t(b)))


def vdot(a, b):
    return _wrap(torch.dot(_t(a).reshape(-1), _t(b).reshape(-1)))


def outer(a, b):
    return _wrap(torch.outer(_t(a).reshape(-1), _t(b).reshape(-1)))


def einsum(spec, *ops, **kw):
    ts = [_t(o) for o in ops]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return _wrap(torch.einsum(spec, *[t.to(dt) for t in ts]))


def tensordot(a, b, axes=2):
    return _wrap(torch.tensordot(_t(a), _t(b), dims=axes))


def diag(x, k=0):
    return _wrap(torch.diag(_t(x), k))


def diagonal(x):
    return _wrap(torch.diagonal(_t(x)))


def trace(x):
    return _wrap(torch.trace(_t(x)))


def transpose(x, axes=None):
    t = _t(x)
    return _wrap(t.permute(*axes) if axes is not None else t.permute(*reversed(range(t.ndim))))


def swapaxes(x, a, b):
    return _wrap(torch.swapaxes(_t(x), a, b))


def reshape(x, shape):
    return _wrap(_t(x).reshape(_shape(shape) if not isinstance(shape, int) else (shape,)))


def ravel(x):
    return _wrap(_t(x).reshape(-1))


def squeeze(x, axis=None):
    return _wrap(_t(x).squeeze() if axis is None else _t(x).squeeze(axis))


def expand_dims(x, axis):
    return _wrap(_t(x).unsqueeze(axis))


def atleast_1d(x):
    return _wrap(torch.atleast_1d(_t(x)))


def atleast_2d(x):
    return _wrap(torch.atleast_2d(_t(x)))


def concatenate(xs, axis=0):
    return _wrap(torch.cat([torch.atleast_1d(_t(v)) for v in xs], dim=axis))


def stack(xs, axis=0):
    ts = [_t(v) for v in xs]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return _wrap(torch.stack([t.to(dt) for t in ts], dim=axis))


def broadcast_to(x, shape):
    return _wrap(torch.broadcast_to(_t(x), _shape(shape)))


def tile(x, reps):
    return _wrap(torch.tile(_t(x), _shape(reps)))


def roll(x, shift, axis=None):
    return _wrap(torch.roll(_t(x), int(shift), None if axis is None else int(axis)))


def flip(x, axis=None):
    t = _t(x)
    return _wrap(torch.flip(t, dims=list(range(t.ndim)) if axis is None else [int(axis)]))


def sort(x, axis=-1):
    return _wrap(torch.sort(_t(x), dim=axis).values)


def argsort(x, axis=-1):
    return _wrap(torch.argsort(_t(x), dim=axis, stable=True))


def take(x, idx, axis=None):
    t = _t(x)
    return _wrap(t.reshape(-1)[_t(idx).long()] if axis is None else torch.index_select(t, axis, _t(idx).long().reshape(-1)))


def tril(x, k=0):
    return _wrap(torch.tril(_t(x), k))


def triu(x, k=0):
    return _wrap(torch.triu(_t(x), k))


def ndim(x):
    return asarray(x).ndim


def shape(x):
    return tuple(asarray(x).shape)


def size(x):
    return int(asarray(x).numel())


def result_type(*xs):
    dt = None
    for x in xs:
        d = x if isinstance(x, torch.dtype) else asarray(x).dtype
        dt = d if dt is None else torch.promote_types(dt, d)
    return dt


def issubdtype(dt, kind):
    dt = _np_dtype_to_torch(dt) if not isinstance(dt, torch.dtype) else dt
    if kind in (_np.floating, floating):
        return dt.is_floating_point
    if kind in (_np.integer, integer):
        return dt in (torch.int8, torch.int16, torch.int32, torch.int64, torch.uint8)
    return False


def finfo(dt):
    return torch.finfo(_np_dtype_to_torch(dt) if not isinstance(dt, torch.dtype) else dt)


def iinfo(dt):
    return torch.iinfo(_np_dtype_to_torch(dt) if not isinstance(dt, torch.dtype) else dt)


def nan_to_num(x, nan=0.0, posinf=None, neginf=None):
    return _wrap(torch.nan_to_num(_t(x), nan=nan, posinf=posinf, neginf=neginf))


def allclose(a, b, rtol=1e-5, atol=1e-8):
    return bool(torch.allclose(_t(a), _t(b), rtol=rtol, atol=atol))


def array_equal(a, b):
    return bool(torch.equal(_t(a), _t(b)))


def float_(x):
    return asarray(x, torch.float32)


# ---- jnp.linalg
linalg = types.ModuleType("jax.numpy.linalg")
linalg.norm = lambda x, ord=None, axis=None: _wrap(torch.linalg.norm(_t(x), ord=ord, dim=axis))
linalg.solve = lambda a, b: _wrap(torch.linalg.solve(_t(a), _t(b)))
linalg.inv = lambda a: _wrap(torch.linalg.inv(_t(a)))
linalg.cholesky = lambda a: _wrap(torch.linalg.cholesky(_t(a)))
linalg.eigh = lambda a: tuple(_wrap(v) for v in torch.linalg.eigh(_t(a)))
linalg.qr = lambda a: tuple(_wrap(v) for v in torch.linalg.qr(_t(a)))
linalg.det = lambda a: _wrap(torch.linalg.det(_t(a)))
linalg.slogdet = lambda a: tuple(_wrap(v) for v in torch.linalg.slogdet(_t(a)))
linalg.__getattr__ = lambda item: _Missing(f"jax.numpy.linalg.{item}")
sys.modules["jax.numpy.linalg"] = linalg

fft = types.ModuleType("jax.numpy.fft")
fft.rfft = lambda x, n=None, axis=-1: _wrap(torch.fft.rfft(_t(x), n=n, dim=axis))
fft.irfft = lambda x, n=None, axis=-1: _wrap(torch.fft.irfft(_t(x), n=n, dim=axis))
fft.fft = lambda x, n=None, axis=-1: _wrap(torch.fft.fft(_t(x), n=n, dim=axis))
fft.ifft = lambda x, n=None, axis=-1: _wrap(torch.fft.ifft(_t(x), n=n, dim=axis))
fft.__getattr__ = lambda item: _Missing(f"jax.numpy.fft.{item}")
sys.modules["jax.numpy.fft"] = fft
