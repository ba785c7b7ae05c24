"""NOT JAX.  A stand-in for the small part of the ``jax`` API that BlackJAX's HMC / NUTS / window-adaptation code
executes, built on torch CPU tensors (float32 default dtype, torch.autograd for ``grad`` / ``value_and_grad``) and
Python control flow -- TEST INFRASTRUCTURE of blackjax_amd, used by ``tests/golden/gen_ref_shim_fixtures.py`` only.

Why: JAX cannot be installed in the build container (no wheel, no network, Python 3.10), so the reference -- pure
Python on JAX -- cannot run there.  With this stand-in on ``sys.path`` (in the generator's process ONLY: nothing else
must ever find a module called ``jax`` here) ``import blackjax`` executes the reference's OWN source from
``/root/reference``: its control flow, tree building, U-turn checkpointing, progressive sampling, Metropolis step,
dual averaging, Welford and window schedule are the reference's code, not a restatement.  What is NOT the reference:

* array arithmetic is torch's fp32 CPU kernels, not XLA:CPU (elementwise results agree to rounding; reductions may
  differ in their last bits: the comparison with the oracle is at a stated tolerance, as it would be with real JAX);
* ``jax.random`` is ``oracle/prng.py`` -- the threefry restatement the engine's kernels are checked against.  The
  random BIT STREAMS are therefore NOT pinned by this (SURVEY row a34 stays "parity unpinned"); what IS pinned is how
  the reference consumes them (which key is split / folded where, which draw decides what);
* ``vmap`` is a Python loop over the leading axis, ``jit`` the identity, ``lax.cond / while_loop / scan / fori_loop``
  Python control flow.

Anything not implemented resolves to a permissive placeholder so that importing the whole ``blackjax`` package works;
calling a placeholder raises."""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np
import torch

torch.set_grad_enabled(False)  # gradients are taken explicitly in value_and_grad / grad only
torch.set_default_dtype(torch.float32)

__version__ = "0.0-refshim"
_THIS = sys.modules[__name__]


# ----------------------------------------------------------------------------------------------- placeholders
class _Missing:
    """Placeholder for API this stand-in does not implement: importable, usable as a decorator at import time,
    fails when a result is actually needed."""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Missing(f"{self._name}.{item}")

    def __call__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs and callable(args[0]) and not isinstance(args[0], _Missing):
            return args[0]  # used as a bare decorator
        if args or kwargs:
            # decorator factory (e.g. partial(jax.jit, static_argnums=...)) or a real call: defer the failure
            return _Missing(self._name + "(...)")
        return _Missing(self._name + "()")

    def __mro_entries__(self, bases):
        return (object,)

    def __getitem__(self, item):
        return self

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __iter__(self):
        raise NotImplementedError(f"refshim: {self._name} is not implemented")

    def __bool__(self):
        raise NotImplementedError(f"refshim: {self._name} is not implemented")

    def __array__(self, *a, **k):
        raise NotImplementedError(f"refshim: {self._name} is not implemented")

    def __repr__(self):
        return f"<refshim placeholder {self._name}>"


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Missing(f"{self.__name__}.{item}")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Any ``jax.<something>`` not defined below, and the third-party packages the reference imports at module level
    but the hot path never calls (optax ...), import as permissive stub modules."""

    ROOTS = ("jax", "fastprogress", "jaxopt", "jaxlib")  # (optax, chex, absl: small restatements beside this package)

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in self.ROOTS and fullname not in sys.modules and fullname != "jax":
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_StubFinder())


def __getattr__(item):
    if item.startswith("__") and item.endswith("__"):
        raise AttributeError(item)
    return _Missing(f"jax.{item}")


# ----------------------------------------------------------------------------------------------- arrays
class Array(torch.Tensor):
    """torch tensor with the few jax.Array / numpy methods the reference uses."""

    @property
    def at(self):
        return _At(self)

    def astype(self, dtype):
        return self.to(_np_dtype_to_torch(dtype))

    @property
    def size(self):  # numpy semantics (number of elements); torch's .size() is a method
        return _SizeProxy(self)

    def item(self):
        return torch.Tensor.item(self)

    def __array__(self, dtype=None, copy=None):
        a = self.detach().as_subclass(torch.Tensor).numpy()
        return a.astype(dtype) if dtype is not None else a

    def block_until_ready(self):
        return self

    def squeeze(self, axis=None):
        t = self.as_subclass(torch.Tensor)
        return _wrap(t.squeeze() if axis is None else t.squeeze(axis))

    def __hash__(self):
        return id(self)

    def __getitem__(self, idx):
        # JAX gather semantics: out-of-bounds integer-array indices are CLAMPED (diagnostics.py:275,297 rely on it)
        adv = _advanced(idx)
        t = self.as_subclass(torch.Tensor)
        if adv is None:
            return _wrap(torch.Tensor.__getitem__(t, _idx(idx) if not isinstance(idx, torch.Tensor) else idx))
        adv = tuple(torch.clamp(torch.where(i < 0, i + t.shape[d], i), 0, t.shape[d] - 1) for d, i in enumerate(adv))
        return _wrap(torch.Tensor.__getitem__(t, adv))

    # booleans take part in arithmetic as 0 / 1 in JAX (``1 - do_accept``, proposal.py:255); torch refuses ``-`` on them
    def _num(self):
        t = self.as_subclass(torch.Tensor)
        return t.to(torch.int32) if t.dtype == torch.bool else t

    def __sub__(self, other):
        o = other._num() if isinstance(other, Array) else other
        return _wrap(self._num() - o)

    def __rsub__(self, other):
        o = other._num() if isinstance(other, Array) else other
        return _wrap(o - self._num())

    def __neg__(self):
        return _wrap(-self._num())

    # NumPy-style reductions (``x.var(axis=0, ddof=1, keepdims=True)``, diagnostics.py:233); internal code works on
    # plain tensors (``_t``), so these only serve the reference's calls
    def _jnp(self):
        return sys.modules["jax.numpy"]

    def sum(self, axis=None, keepdims=False, dtype=None, dim=None, keepdim=None, out=None):
        return self._jnp().sum(self, axis if dim is None else dim, keepdims if keepdim is None else keepdim)

    def mean(self, axis=None, keepdims=False, dim=None, keepdim=None, dtype=None, out=None):  # (np.mean(x) calls x.mean(dtype=, out=))
        return self._jnp().mean(self, axis if dim is None else dim, keepdims if keepdim is None else keepdim)

    def var(self, axis=None, ddof=0, keepdims=False, dtype=None, out=None):
        return self._jnp().var(self, axis, ddof, keepdims)

    def std(self, axis=None, ddof=0, keepdims=False, dtype=None, out=None):
        return self._jnp().std(self, axis, ddof, keepdims)

    def max(self, axis=None, keepdims=False):
        return self._jnp().max(self, axis, keepdims)

    def min(self, axis=None, keepdims=False):
        return self._jnp().min(self, axis, keepdims)

    def any(self, axis=None, keepdims=False):
        return self._jnp().any(self, axis, keepdims)

    def all(self, axis=None, keepdims=False):
        return self._jnp().all(self, axis, keepdims)

    def prod(self, axis=None):
        return self._jnp().prod(self, axis)

    def argsort(self, axis=-1):
        return self._jnp().argsort(self, axis)

    def ravel(self):
        return self._jnp().ravel(self)

    def flatten(self):
        return self._jnp().ravel(self)

    def swapaxes(self, a, b):
        return self._jnp().swapaxes(self, a, b)

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        return self._jnp().transpose(self, axes or None)

    def copy(self):
        return self.clone()

    def conj(self):
        return self._jnp().conjugate(self)


class _SizeProxy(int):
    """``x.size`` as an int that can still be called like torch's ``x.size()`` / ``x.size(0)``."""

    def __new__(cls, t):
        obj = super().__new__(cls, t.numel())
        obj._t = t
        return obj

    def __call__(self, *a):
        return torch.Tensor.size(self._t, *a)


class _At:
    def __init__(self, x):
        self.x = x

    def __getitem__(self, idx):
        return _AtIdx(self.x, idx)


def _idx(idx):
    if isinstance(idx, tuple):
        return tuple(_idx(i) for i in idx)
    if isinstance(idx, torch.Tensor) and idx.ndim == 0:
        return int(idx)
    return idx


def _advanced(idx):
    """(tensors, ok): ``idx`` as a tuple of index tensors when it consists of integer tensors / ints only."""
    parts = idx if isinstance(idx, tuple) else (idx,)
    if not parts or not all(isinstance(p, (int, torch.Tensor)) and not (isinstance(p, torch.Tensor) and p.dtype == torch.bool)
                            for p in parts):
        return None
    if not any(isinstance(p, torch.Tensor) for p in parts):
        return None
    return tuple(p.as_subclass(torch.Tensor).long() if isinstance(p, torch.Tensor) else torch.tensor(p) for p in parts)


class _AtIdx:
    def __init__(self, x, idx):
        self.x, self.idx = x, idx

    def set(self, v):
        y = self.x.clone()
        adv = _advanced(self.idx)
        if adv is None:
            y[_idx(self.idx)] = v
            return y
        # JAX scatter semantics: updates at out-of-bounds indices are DROPPED (diagnostics.py:275 relies on it)
        adv = torch.broadcast_tensors(*adv)
        ok = torch.ones_like(adv[0], dtype=torch.bool)
        for d, i in enumerate(adv):
            ok &= (i >= -y.shape[d]) & (i < y.shape[d])
        vt = torch.broadcast_to(asarray(v).as_subclass(torch.Tensor).to(y.dtype), adv[0].shape + tuple(y.shape[len(adv):]))
        yt = y.as_subclass(torch.Tensor)
        yt[tuple(i[ok] for i in adv)] = vt[ok]
        return y

    def add(self, v):
        y = self.x.clone()
        y[_idx(self.idx)] += v
        return y

    def multiply(self, v):
        y = self.x.clone()
        y[_idx(self.idx)] *= v
        return y

    def get(self):
        return self.x[self.idx]  # (Array.__getitem__: clamped like a JAX gather)


_DT = {np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64,
       np.bool_: torch.bool, np.uint32: torch.int64, float: torch.float32, int: torch.int32, bool: torch.bool,
       "float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "int64": torch.int64, "bool": torch.bool}


def _np_dtype_to_torch(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    if dtype in _DT:
        return _DT[dtype]
    return _DT[np.dtype(dtype).type]


def _wrap(x):
    if isinstance(x, torch.Tensor) and not isinstance(x, Array):
        return x.as_subclass(Array)
    return x


def asarray(x, dtype=None):
    """jnp.asarray: Python floats -> float32, Python ints -> int32 (JAX's defaults with x64 off)."""
    dtype = _np_dtype_to_torch(dtype)
    if isinstance(x, torch.Tensor):
        return _wrap(x if dtype is None else x.to(dtype))
    if isinstance(x, np.ndarray) or isinstance(x, np.generic):
        a = np.asarray(x)
        if a.dtype == np.float64 and dtype is None:
            a = a.astype(np.float32)
        if a.dtype == np.uint32:
            a = a.astype(np.int64)
        t = torch.from_numpy(np.array(a, copy=True, order="C"))  # (ascontiguousarray would turn 0-d into 1-d)
        return _wrap(t if dtype is None else t.to(dtype))
    if isinstance(x, (list, tuple)) and any(isinstance(v, torch.Tensor) for v in x):
        return _wrap(torch.stack([asarray(v, dtype) for v in x]))
    if isinstance(x, bool):
        return _wrap(torch.tensor(x, dtype=dtype or torch.bool))
    if isinstance(x, int):
        return _wrap(torch.tensor(x, dtype=dtype or torch.int32))
    if isinstance(x, float):
        return _wrap(torch.tensor(x, dtype=dtype or torch.float32))
    a = np.asarray(x)
    return asarray(a, dtype)


# ----------------------------------------------------------------------------------------------- pytrees
def _is_namedtuple(x):
    return isinstance(x, tuple) and hasattr(x, "_fields")


def tree_flatten(tree):
    """-> (leaves, treedef); ``None`` is an empty node (as in JAX), dict keys are visited sorted."""
    leaves = []

    def rec(t):
        if t is None:
            return ("none",)
        if _is_namedtuple(t):
            return ("nt", type(t), [rec(v) for v in t])
        if isinstance(t, tuple):
            return ("tuple", [rec(v) for v in t])
        if isinstance(t, list):
            return ("list", [rec(v) for v in t])
        if isinstance(t, dict):
            keys = sorted(t)
            return ("dict", keys, [rec(t[k]) for k in keys])
        leaves.append(t)
        return ("leaf",)

    return leaves, rec(tree)


def tree_unflatten(treedef, leaves):
    it = iter(leaves)

    def rec(d):
        kind = d[0]
        if kind == "none":
            return None
        if kind == "leaf":
            return next(it)
        if kind == "nt":
            return d[1](*[rec(c) for c in d[2]])
        if kind == "tuple":
            return tuple(rec(c) for c in d[1])
        if kind == "list":
            return [rec(c) for c in d[1]]
        return {k: rec(c) for k, c in zip(d[1], d[2])}

    return rec(treedef)


def _flatten_up_to(treedef, tree):
    """The parts of ``tree`` that sit where ``treedef`` has its leaves (``tree`` has ``treedef`` as a PREFIX: a leaf of the
    first tree may face a whole subtree of the others -- adaptation/base.py:51 maps field NAMES over a tuple of pytrees)."""
    out = []

    def rec(d, t):
        kind = d[0]
        if kind == "leaf":
            out.append(t)
        elif kind == "none":
            assert t is None, "tree_map: trees do not match"
        elif kind in ("nt", "tuple", "list"):
            children = d[2] if kind == "nt" else d[1]
            assert isinstance(t, (tuple, list)) and len(t) == len(children), "tree_map: trees do not match"
            for c, v in zip(children, t):
                rec(c, v)
        else:
            assert isinstance(t, dict) and sorted(t) == d[1], "tree_map: trees do not match"
            for k, c in zip(d[1], d[2]):
                rec(c, t[k])

    rec(treedef, tree)
    return out


def tree_map(f, tree, *rest, is_leaf=None):
    leaves, treedef = tree_flatten(tree)
    others = [_flatten_up_to(treedef, r) for r in rest]
    return tree_unflatten(treedef, [f(*xs) for xs in zip(leaves, *others)])


def tree_leaves(tree, is_leaf=None):
    return tree_flatten(tree)[0]


def tree_structure(tree):
    return tree_flatten(tree)[1]


tree = types.ModuleType("jax.tree")
tree.map, tree.leaves, tree.flatten, tree.unflatten, tree.structure = tree_map, tree_leaves, tree_flatten, tree_unflatten, tree_structure
tree.reduce = lambda f, t, initializer=None: __import__("functools").reduce(f, tree_leaves(t)) if initializer is None else __import__("functools").reduce(f, tree_leaves(t), initializer)
tree_util = types.ModuleType("jax.tree_util")
tree_util.tree_map, tree_util.tree_leaves, tree_util.tree_flatten, tree_util.tree_unflatten = tree_map, tree_leaves, tree_flatten, tree_unflatten
tree_util.tree_structure = tree_structure
tree_util.tree_reduce = tree.reduce
tree_util.Partial = __import__("functools").partial
tree_util.register_pytree_node_class = lambda cls: cls
tree_util.register_pytree_node = lambda *a, **k: None
tree_util.register_dataclass = lambda cls, *a, **k: cls
tree_util.__getattr__ = lambda item: _Missing(f"jax.tree_util.{item}")
sys.modules["jax.tree"], sys.modules["jax.tree_util"] = tree, tree_util

flatten_util = types.ModuleType("jax.flatten_util")


def ravel_pytree(pytree):
    leaves, treedef = tree_flatten(pytree)
    leaves = [asarray(v) for v in leaves]
    shapes = [tuple(v.shape) for v in leaves]
    sizes = [int(v.numel()) for v in leaves]
    flat = _wrap(torch.cat([v.reshape(-1) for v in leaves])) if leaves else asarray(np.zeros(0, np.float32))

    def unravel(x):
        parts = torch.split(x, sizes) if sizes else []
        return tree_unflatten(treedef, [_wrap(p.reshape(s)) for p, s in zip(parts, shapes)])

    return flat, unravel


flatten_util.ravel_pytree = ravel_pytree
sys.modules["jax.flatten_util"] = flatten_util
_src = _StubModule("jax._src")
_src.__path__ = []
_src_fu = types.ModuleType("jax._src.flatten_util")
_src_fu.ravel_pytree = ravel_pytree
sys.modules["jax._src"], sys.modules["jax._src.flatten_util"] = _src, _src_fu


# ----------------------------------------------------------------------------------------------- transforms
def jit(f=None, static_argnums=(), static_argnames=(), **kwargs):
    """The identity, except that -- as under a real ``jit`` -- Python scalars passed in non-static positions arrive as
    arrays (``util.linear_map(1.0, x)`` reads ``.dtype`` of its first argument)."""
    if f is None:
        return lambda g: jit(g, static_argnums=static_argnums, static_argnames=static_argnames)
    import functools

    nums = (static_argnums,) if isinstance(static_argnums, int) else tuple(static_argnums or ())
    names = (static_argnames,) if isinstance(static_argnames, str) else tuple(static_argnames or ())

    def conv(v):
        return tree_map(lambda x: asarray(x) if isinstance(x, (bool, int, float)) else x, v)

    @functools.wraps(f)
    def wrapped(*args, **kw):
        args = tuple(a if i in nums else conv(a) for i, a in enumerate(args))
        kw = {k: (v if k in names else conv(v)) for k, v in kw.items()}
        return f(*args, **kw)

    return wrapped


def _n_batch(leaves, in_axes_leaves):
    for v, ax in zip(leaves, in_axes_leaves):
        if ax is not None:
            return v.shape[0] if not isinstance(v, np.ndarray) else v.shape[0]
    raise ValueError("vmap: nothing to map over")


def _stack(items):
    first = items[0]
    if isinstance(first, torch.Tensor):
        return _wrap(torch.stack([t.as_subclass(torch.Tensor) for t in items]))
    if isinstance(first, np.ndarray):  # keys
        return np.stack(items)
    if isinstance(first, (bool, int, float, np.generic)):
        return asarray(np.asarray(items))
    raise TypeError(f"vmap: cannot stack outputs of type {type(first)}")


def vmap(f, in_axes=0, out_axes=0, **kwargs):
    """A Python loop over the leading axis of every mapped leaf (``in_axes``: 0 / None per argument, or one value)."""
    if out_axes != 0:
        raise NotImplementedError("refshim vmap: out_axes != 0")

    def mapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        flat, defs, ax_flat = [], [], []
        for a, ax in zip(args, axes):
            leaves, d = tree_flatten(a)
            if isinstance(ax, (tuple, list, dict)) or _is_namedtuple(ax):
                ax_leaves = []
                for sub_a, sub_ax in zip(a if not isinstance(a, dict) else [a[k] for k in sorted(a)],
                                         ax if not isinstance(ax, dict) else [ax[k] for k in sorted(ax)]):
                    ax_leaves += [sub_ax] * len(tree_flatten(sub_a)[0])
            else:
                if ax not in (0, None):
                    raise NotImplementedError("refshim vmap: only axis 0 / None")
                ax_leaves = [ax] * len(leaves)
            flat.append(leaves)
            defs.append(d)
            ax_flat.append(ax_leaves)
        n = _n_batch([v for ls in flat for v in ls], [a for ls in ax_flat for a in ls])
        outs = []
        for i in range(n):
            call = [tree_unflatten(d, [v if ax is None else v[i] for v, ax in zip(ls, axs)])
                    for ls, d, axs in zip(flat, defs, ax_flat)]
            outs.append(f(*call))
        out_leaves = [tree_flatten(o)[0] for o in outs]
        out_def = tree_flatten(outs[0])[1]
        return tree_unflatten(out_def, [_stack([ol[j] for ol in out_leaves]) for j in range(len(out_leaves[0]))])

    return mapped


def value_and_grad(fun, argnums=0, has_aux=False, **kwargs):
    nums = (argnums,) if isinstance(argnums, int) else tuple(argnums)

    def vg(*args, **kw):
        flats = [tree_flatten(args[i]) for i in nums]
        with torch.enable_grad():
            reqs = [[asarray(v).detach().clone().as_subclass(torch.Tensor).requires_grad_(True) for v in leaves]
                    for leaves, _ in flats]
            call = list(args)
            for i, (_, d), req in zip(nums, flats, reqs):
                call[i] = tree_unflatten(d, [_wrap(r) for r in req])
            out = fun(*call, **kw)
            val, aux = out if has_aux else (out, None)
            val_t = asarray(val)
            flat_req = [r for req in reqs for r in req]
            if val_t.requires_grad:
                grads = torch.autograd.grad(val_t.as_subclass(torch.Tensor), flat_req, allow_unused=True)
            else:
                grads = [None] * len(flat_req)
        grads = [_wrap(torch.zeros_like(r) if g is None else g.detach()) for g, r in zip(grads, flat_req)]
        trees, k = [], 0
        for (_, d), req in zip(flats, reqs):
            trees.append(tree_unflatten(d, grads[k:k + len(req)]))
            k += len(req)
        g_out = trees[0] if isinstance(argnums, int) else tuple(trees)
        val_d = _wrap(val_t.detach())
        return ((val_d, aux), g_out) if has_aux else (val_d, g_out)

    return vg


def grad(fun, argnums=0, has_aux=False, **kwargs):
    vg = value_and_grad(fun, argnums, has_aux)

    def g(*args, **kw):
        out = vg(*args, **kw)
        return (out[1], out[0][1]) if has_aux else out[1]

    return g


def device_put(x, *a, **k):
    return x


def block_until_ready(x):
    return x


def devices(*a, **k):
    return ["refshim-cpu"]


def device_count(*a, **k):
    return 1


def local_device_count(*a, **k):
    return 1


class _Config:
    jax_threefry_partitionable = True
    jax_enable_x64 = False

    def update(self, *a, **k):
        pass

    def __getattr__(self, item):
        return None


config = _Config()

# ----------------------------------------------------------------------------------------------- submodules
# (explicit imports: with a module-level __getattr__ in place ``from . import lax`` would find the placeholder)
import importlib as _importlib  # noqa: E402

numpy = _importlib.import_module(__name__ + ".numpy")
lax = _importlib.import_module(__name__ + ".lax")
random = _importlib.import_module(__name__ + ".random")
scipy = _importlib.import_module(__name__ + ".scipy")

nn = _StubModule("jax.nn")
nn.logsumexp = lambda a, axis=None, b=None, keepdims=False: scipy.special.logsumexp(a, axis=axis, keepdims=keepdims)
nn.sigmoid = lambda x: scipy.special.expit(x)
nn.softmax = lambda x, axis=-1: _wrap(torch.softmax(asarray(x).as_subclass(torch.Tensor), dim=axis))
sys.modules["jax.nn"] = nn

typing = _StubModule("jax.typing")
typing.ArrayLike = object
sys.modules["jax.typing"] = typing
