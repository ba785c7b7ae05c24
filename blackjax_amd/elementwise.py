"""``from_elementwise``: a plain PyTorch log-density -> ONE HIP value-and-gradient kernel (external callable).

The reference's user writes ``logdensity_fn`` as array code and ``jax.value_and_grad`` + XLA fuse it
(blackjax/mcmc/integrators.py:189,204).  A PyTorch callable evaluated eagerly under autograd moves 50-80 bytes per
element (one pass per op, forward and backward: INTEGRATION.md section 4) where the sampler's roofline assumes 8
(read q, write grad).  For the common shape of a batched log-density --

    logp(q)[n] = c + sum_t a_t * sum_j f_t(q[n, j], theta_1[j], theta_2[j], ...)

i.e. element-wise arithmetic on ``q: (N, D)``, ``(D,)`` parameter tensors and scalars, reduced over the last axis,
the row sums combined linearly -- this module traces the function with ``torch.fx``, differentiates every
element-wise node in forward mode (d/dq[n, j]: the Jacobian is diagonal), and emits the HIP source of a
``blackjax_amd.targets.DeviceTarget``: one wave per chain row, the row in registers, ``grad`` written as it is
computed, the row sums accumulated in fp64 and reduced with the wave's DPP tree.  hiprtc compiles it on first use.

The result is an ordinary callable ``f(q) -> (logp, grad)`` under the engine's external-callable contract (it is
NOT evaluated inside the engine's kernels unless the user also passes ``fuse_target=True``).  Numerics: each traced
op is one fp32 operation (``-ffp-contract=off``), as in eager PyTorch; the gradient is the forward-mode formula, not
autograd's reverse sweep, and the row sum is fp64-accumulated, so values agree with autograd to rounding
(tests/test_elementwise_gpu.py states the tolerance), not bit for bit.

Anything outside that shape raises ``NotImplementedError`` with the offending node: pass the plain callable then.
"""
from __future__ import annotations

import math
import operator
from typing import Callable

import torch

__all__ = ["from_elementwise", "ElementwiseSource"]


class _E:
    """Element-wise value: C expressions of the value and of its derivative w.r.t. this element of q
    (``d is None``: does not depend on q)."""

    __slots__ = ("v", "d")

    def __init__(self, v: str, d):
        self.v, self.d = v, d


class _R:
    """Row value (after a reduction over the last axis): ``const + sum_t coef[t] * term_t``."""

    __slots__ = ("coef", "const")

    def __init__(self, coef: dict, const: float = 0.0):
        self.coef, self.const = coef, float(const)


def _lit(x: float) -> str:
    x = float(x)
    if math.isinf(x) or math.isnan(x):
        raise NotImplementedError("from_elementwise: non-finite constant")
    import numpy as np

    return f"{float(np.float32(x))!r}f".replace("e+", "e")


class ElementwiseSource:
    """The generated device source + its parameter table (``params``: ``(n_params, D)`` float32)."""

    def __init__(self, source: str, params, n_terms: int, description: str):
        self.source, self.params, self.n_terms, self.description = source, params, n_terms, description


class _Gen:
    def __init__(self, dim: int, device):
        self.D, self.device = int(dim), device
        self.lines: list = []
        self.n = 0
        self.params: list = []  # (D,) float32 tensors
        self.terms: list = []  # element-wise node of each row sum

    def tmp(self, expr: str) -> str:
        name = f"t{self.n}"
        self.n += 1
        self.lines.append(f"const float {name} = {expr};")
        return name

    # ---- leaves
    def const(self, x: float) -> _E:
        return _E(_lit(x), None)

    def param(self, t: torch.Tensor) -> _E:
        t = t.detach()
        if t.numel() == 1:
            return self.const(float(t))
        if tuple(t.shape) not in ((self.D,), (1, self.D)):
            raise NotImplementedError(f"from_elementwise: a captured tensor of shape {tuple(t.shape)} (only scalars "
                                      f"and ({self.D},) vectors broadcast over the chains are supported)")
        self.params.append(t.reshape(self.D).to(device=self.device, dtype=torch.float32))
        return _E(f"P{len(self.params) - 1}", None)

    # ---- arithmetic (every op one rounded fp32 operation, as eager PyTorch)
    def add(self, a: _E, b: _E, sign: str = "+") -> _E:
        v = self.tmp(f"{a.v} {sign} {b.v}")
        if a.d is None and b.d is None:
            return _E(v, None)
        if b.d is None:
            return _E(v, a.d)
        if a.d is None:
            return _E(v, b.d if sign == "+" else self.tmp(f"-{b.d}"))
        return _E(v, self.tmp(f"{a.d} {sign} {b.d}"))

    def neg(self, a: _E) -> _E:
        return _E(self.tmp(f"-{a.v}"), None if a.d is None else self.tmp(f"-{a.d}"))

    def mul(self, a: _E, b: _E) -> _E:
        v = self.tmp(f"{a.v} * {b.v}")
        if a.d is None and b.d is None:
            return _E(v, None)
        if b.d is None:
            return _E(v, self.tmp(f"{a.d} * {b.v}"))
        if a.d is None:
            return _E(v, self.tmp(f"{a.v} * {b.d}"))
        if a.v == b.v and a.d == b.d:  # x * x
            return _E(v, self.tmp(f"2.0f * ({a.v} * {a.d})"))
        return _E(v, self.tmp(f"{a.d} * {b.v} + {a.v} * {b.d}"))

    def div(self, a: _E, b: _E) -> _E:
        v = self.tmp(f"{a.v} / {b.v}")
        if a.d is None and b.d is None:
            return _E(v, None)
        if b.d is None:
            return _E(v, self.tmp(f"{a.d} / {b.v}"))
        if a.d is None:
            return _E(v, self.tmp(f"-({v} * {b.d}) / {b.v}"))
        return _E(v, self.tmp(f"({a.d} - {v} * {b.d}) / {b.v}"))

    def powc(self, a: _E, n: float) -> _E:
        n = float(n)
        if n == 2.0:
            return self.mul(a, a)
        if n == 1.0:
            return a
        if n == 0.5:
            return self.unary("sqrt", a)
        if n == -1.0:
            return self.div(self.const(1.0), a)
        if n == 3.0:
            sq = self.mul(a, a)
            return self.mul(sq, a)
        v = self.tmp(f"powf({a.v}, {_lit(n)})")
        if a.d is None:
            return _E(v, None)
        return _E(v, self.tmp(f"{_lit(n)} * powf({a.v}, {_lit(n - 1.0)}) * {a.d}"))

    def unary(self, name: str, a: _E) -> _E:
        av = a.v
        dd = (lambda f: None if a.d is None else self.tmp(f.replace("DA", a.d)))
        if name == "exp":
            v = self.tmp(f"expf({av})")
            return _E(v, dd(f"{v} * DA"))
        if name == "log":
            return _E(self.tmp(f"logf({av})"), dd(f"DA / {av}"))
        if name == "log1p":
            return _E(self.tmp(f"log1pf({av})"), dd(f"DA / (1.0f + {av})"))
        if name == "expm1":
            v = self.tmp(f"expm1f({av})")
            return _E(v, dd(f"({v} + 1.0f) * DA"))
        if name == "sqrt":
            v = self.tmp(f"sqrtf({av})")
            return _E(v, dd(f"DA / (2.0f * {v})"))
        if name == "rsqrt":
            v = self.tmp(f"(1.0f / sqrtf({av}))")
            return _E(v, dd(f"-0.5f * {v} / {av} * DA"))
        if name == "tanh":
            v = self.tmp(f"tanhf({av})")
            return _E(v, dd(f"(1.0f - {v} * {v}) * DA"))
        if name == "sigmoid":
            v = self.tmp(f"(1.0f / (1.0f + expf(-{av})))")
            return _E(v, dd(f"{v} * (1.0f - {v}) * DA"))
        if name == "logsigmoid":  # -softplus(-a)
            v = self.tmp(f"(({av}) < -20.0f ? ({av}) : -log1pf(expf(-({av}))))")
            return _E(v, dd(f"(1.0f / (1.0f + expf({av}))) * DA"))
        if name == "softplus":  # torch defaults: beta = 1, threshold = 20
            v = self.tmp(f"(({av}) > 20.0f ? ({av}) : log1pf(expf({av})))")
            return _E(v, dd(f"(1.0f / (1.0f + expf(-({av})))) * DA"))
        if name == "sin":
            return _E(self.tmp(f"sinf({av})"), dd(f"cosf({av}) * DA"))
        if name == "cos":
            return _E(self.tmp(f"cosf({av})"), dd(f"-sinf({av}) * DA"))
        if name == "abs":
            return _E(self.tmp(f"fabsf({av})"), dd(f"(({av}) > 0.0f ? 1.0f : (({av}) < 0.0f ? -1.0f : 0.0f)) * DA"))
        if name == "reciprocal":
            return self.div(self.const(1.0), a)
        if name == "square":
            return self.mul(a, a)
        if name == "lgamma":
            # d/dx lgamma = digamma: not in the device math library -> central difference would not be exact; refuse
            raise NotImplementedError("from_elementwise: lgamma of a q-dependent value (no device digamma)")
        raise NotImplementedError(f"from_elementwise: unsupported element-wise function {name!r}")


_BIN = {operator.add: "add", torch.add: "add", operator.sub: "sub", torch.sub: "sub", operator.mul: "mul",
        torch.mul: "mul", operator.truediv: "div", torch.div: "div", torch.true_divide: "div",
        operator.pow: "pow", torch.pow: "pow"}
_UN = {torch.exp: "exp", torch.log: "log", torch.log1p: "log1p", torch.expm1: "expm1", torch.sqrt: "sqrt",
       torch.rsqrt: "rsqrt", torch.tanh: "tanh", torch.sigmoid: "sigmoid", torch.sin: "sin", torch.cos: "cos",
       torch.abs: "abs", operator.abs: "abs", torch.reciprocal: "reciprocal", torch.square: "square",
       torch.nn.functional.softplus: "softplus", torch.nn.functional.logsigmoid: "logsigmoid",
       torch.nn.functional.sigmoid: "sigmoid", torch.nn.functional.tanh: "tanh", torch.lgamma: "lgamma"}
_UN_METHODS = {"exp", "log", "log1p", "expm1", "sqrt", "rsqrt", "tanh", "sigmoid", "sin", "cos", "abs",
               "reciprocal", "square", "lgamma"}


def _is_last_axis(args, kwargs) -> bool:
    """``sum(dim[, keepdim])`` arguments (positional or by name) of a row sum this generator can emit: the last axis of
    the (N, D) batch, keepdim false, no ``dtype=``.  Anything else is refused (the caller raises NotImplementedError):
    ``q.sum(-1, True)`` is (N, 1) in PyTorch, not the (N,) row sum."""
    if len(args) > 2 or set(kwargs) - {"dim", "axis", "keepdim"}:  # extra positionals, dtype=, out= ...
        return False
    dim = args[0] if args else kwargs.get("dim", kwargs.get("axis"))
    keepdim = args[1] if len(args) > 1 else kwargs.get("keepdim", False)
    if not isinstance(keepdim, (bool, int)) or keepdim:
        return False
    if isinstance(dim, (tuple, list)) and len(dim) == 1:
        dim = dim[0]
    return isinstance(dim, int) and not isinstance(dim, bool) and dim in (-1, 1)  # the batch is 2-D: (N, D)


def trace(fn: Callable, dim: int, device="cpu") -> ElementwiseSource:
    """``fn`` -> device source of a ``DeviceTarget`` struct (no GPU needed).  See the module docstring."""
    import torch.fx as fx

    try:
        gm = fx.symbolic_trace(fn if isinstance(fn, torch.nn.Module) else _Wrap(fn))
    except Exception as e:  # data-dependent control flow, shape arithmetic on q, ...: not an element-wise expression
        raise NotImplementedError(f"from_elementwise: torch.fx could not trace the function ({type(e).__name__}: {e}); "
                                  "pass the plain callable instead") from e
    gen = _Gen(dim, device)
    env: dict = {}
    out = None

    def val(a):
        if isinstance(a, fx.Node):
            return env[a]
        if isinstance(a, (int, float)):
            return gen.const(float(a))
        if isinstance(a, torch.Tensor):
            return gen.param(a)
        raise NotImplementedError(f"from_elementwise: argument {a!r}")

    def binary(kind, a, b, node):
        ra, rb = isinstance(a, _R), isinstance(b, _R)
        if ra or rb:  # row arithmetic: linear combinations of row sums only
            def scalar(e):
                if isinstance(e, _E) and e.d is None and not e.v.startswith("P"):
                    try:
                        return float(e.v.rstrip("f"))
                    except ValueError:
                        pass
                raise NotImplementedError(f"from_elementwise: node {node.format_node()} combines a row sum with a "
                                          "non-constant value (only linear combinations of row sums are supported)")
            if kind in ("add", "sub"):
                sg = 1.0 if kind == "add" else -1.0
                if ra and rb:
                    co = dict(a.coef)
                    for t, c in b.coef.items():
                        co[t] = co.get(t, 0.0) + sg * c
                    return _R(co, a.const + sg * b.const)
                if ra:
                    return _R(dict(a.coef), a.const + sg * scalar(b))
                return _R({t: sg * c for t, c in b.coef.items()}, scalar(a) + sg * b.const)
            if kind == "mul" and (ra != rb):
                r, s = (a, scalar(b)) if ra else (b, scalar(a))
                return _R({t: c * s for t, c in r.coef.items()}, r.const * s)
            if kind == "div" and ra and not rb:
                s = scalar(b)
                return _R({t: c / s for t, c in a.coef.items()}, a.const / s)
            raise NotImplementedError(f"from_elementwise: node {node.format_node()} is not linear in the row sums")
        if kind == "add":
            return gen.add(a, b, "+")
        if kind == "sub":
            return gen.add(a, b, "-")
        if kind == "mul":
            return gen.mul(a, b)
        if kind == "div":
            return gen.div(a, b)
        if kind == "pow":
            if b.d is None and not b.v.startswith(("P", "t")):
                return gen.powc(a, float(b.v.rstrip("f")))
            raise NotImplementedError("from_elementwise: pow with a non-constant exponent")
        raise NotImplementedError(kind)

    def reduce_sum(e, node):
        if isinstance(e, _R):
            raise NotImplementedError(f"from_elementwise: {node.format_node()} reduces a row value again")
        gen.terms.append(e)
        return _R({len(gen.terms) - 1: 1.0})

    for node in gm.graph.nodes:
        if node.op == "placeholder":
            if env:
                raise NotImplementedError("from_elementwise: the log-density takes ONE argument, q: (N, D)")
            env[node] = _E("x", "1.0f")
        elif node.op == "get_attr":
            obj = gm
            for part in node.target.split("."):
                obj = getattr(obj, part)
            env[node] = gen.param(obj if isinstance(obj, torch.Tensor) else torch.as_tensor(obj))
        elif node.op == "call_function":
            f = node.target
            if f in _BIN:
                a, b = val(node.args[0]), val(node.args[1])
                env[node] = binary(_BIN[f], a, b, node)
            elif f in (operator.neg, torch.neg, torch.negative):
                a = val(node.args[0])
                env[node] = _R({t: -c for t, c in a.coef.items()}, -a.const) if isinstance(a, _R) else gen.neg(a)
            elif f in _UN:
                a = val(node.args[0])
                if isinstance(a, _R):
                    raise NotImplementedError(f"from_elementwise: {node.format_node()} applies a non-linear function to a row sum")
                if _UN[f] == "softplus" and (len(node.args) > 1 or node.kwargs):
                    raise NotImplementedError("from_elementwise: softplus with non-default beta / threshold")
                env[node] = gen.unary(_UN[f], a)
            elif f is torch.sum:
                if not _is_last_axis(node.args[1:], node.kwargs):
                    raise NotImplementedError(f"from_elementwise: {node.format_node()}: only sums over the last axis")
                env[node] = reduce_sum(val(node.args[0]), node)
            else:
                raise NotImplementedError(f"from_elementwise: unsupported function in {node.format_node()}")
        elif node.op == "call_method":
            m = node.target
            a = val(node.args[0])
            if m == "sum":
                if not _is_last_axis(node.args[1:], node.kwargs):
                    raise NotImplementedError(f"from_elementwise: {node.format_node()}: only sums over the last axis")
                env[node] = reduce_sum(a, node)
            elif m in ("add", "sub", "mul", "div", "true_divide", "pow"):
                env[node] = binary({"true_divide": "div"}.get(m, m), a, val(node.args[1]), node)
            elif m == "neg":
                env[node] = _R({t: -c for t, c in a.coef.items()}, -a.const) if isinstance(a, _R) else gen.neg(a)
            elif m in _UN_METHODS:
                if isinstance(a, _R):
                    raise NotImplementedError(f"from_elementwise: {node.format_node()} applies a non-linear function to a row sum")
                env[node] = gen.unary(m, a)
            elif m in ("float", "contiguous", "clone"):
                env[node] = a
            else:
                raise NotImplementedError(f"from_elementwise: unsupported method in {node.format_node()}")
        elif node.op == "output":
            out = val(node.args[0])
        else:
            raise NotImplementedError(f"from_elementwise: {node.format_node()}")
    if not isinstance(out, _R) or not out.coef:
        raise NotImplementedError("from_elementwise: the function must return a sum over the last axis of q "
                                  "(a (N,) tensor of log-densities)")
    used = sorted(out.coef)
    for t in used:
        if gen.terms[t].d is None:
            out.const += 0.0  # a q-independent row sum: contributes to logp, not to the gradient
    n_p = len(gen.params)
    body = "\n          ".join(gen.lines)
    p_decl = "".join(f"const float P{i} = pv{i}[e]; " for i in range(n_p))
    p_load = "".join(f"const F4 pq{i} = ld4(params + {i} * D + j); const float pv{i}[4] = {{pq{i}.x, pq{i}.y, pq{i}.z, pq{i}.w}};\n        "
                     for i in range(n_p))
    g_terms = [f"{_lit(out.coef[t])} * {gen.terms[t].d}" for t in used if gen.terms[t].d is not None]
    g_expr = " + ".join(g_terms) if g_terms else "0.0f"
    acc_decl = "".join(f"double acc{t} = 0.0; " for t in used)
    acc_add = "".join(f"if (need_logp) acc{t} += (double){gen.terms[t].v}; " for t in used)
    lp_expr = " + ".join([repr(float(out.const))] + [f"{float(out.coef[t])!r} * wave_sum(acc{t})" for t in used])
    source = f"""
// generated by blackjax_amd.elementwise.from_elementwise -- {len(gen.lines)} fp32 operations per element, {n_p} (D,) parameter vector(s)
struct Target {{
  template <int NI> struct Ctx {{}};
  template <int NI> static __device__ void init(Ctx<NI>&, int64_t, const float*) {{}}
  template <int NI>
  static __device__ void eval(const Ctx<NI>&, int64_t D, const float* __restrict__ params, const F4 (&xr)[NI],
                              bool need_logp, F4 (&g)[NI], float& lp) {{
    const int lane = threadIdx.x & 63;
    {acc_decl}
#pragma unroll
    for (int k = 0; k < NI; ++k) {{
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {{
        {p_load}const float xs[4] = {{xr[k].x, xr[k].y, xr[k].z, xr[k].w}};
        float gs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {{
          const float x = xs[e]; {p_decl}
          {body}
          gs[e] = {g_expr};
          {acc_add}
        }}
        g[k] = F4{{gs[0], gs[1], gs[2], gs[3]}};
      }}
    }}
    if (need_logp) lp = (float)({lp_expr});
  }}
}};
"""
    # The same element-wise body as a ROW-LOOP kernel for any D (rows longer than 1 024 floats, D % 4 != 0): one wave per
    # chain row, spans of 4 x 1 KiB requested before they are consumed (row_sweep4), 4-byte accesses when D % 4 != 0.  A
    # lane meets its elements in ascending column order, exactly as in `eval` above, so for D <= 1 024, D % 4 == 0 the
    # fp64 partial sums -- and with them logp -- are the struct's bit for bit.  External callable only (the engine's
    # kernels keep a row in registers: `eval`).
    p_load_u = "".join(f"pq{i}[u] = ld4(params + {i} * D + j); " for i in range(n_p))
    p_decl_arr = "".join(f"F4 pq{i}[4]; " for i in range(n_p))
    p_unpack_u = "".join(f"const float pv{i}[4] = {{pq{i}[u].x, pq{i}[u].y, pq{i}[u].z, pq{i}[u].w}}; " for i in range(n_p))
    p_scalar = "".join(f"const float P{i} = params[{i} * D + j]; " for i in range(n_p))
    acc_add_rows = "".join(f"acc{t} += (double){gen.terms[t].v}; " for t in used)
    rows_source = f"""
// generated by blackjax_amd.elementwise.from_elementwise -- row-loop form (any D)
extern "C" __global__ void __launch_bounds__(256) bjx_rtc_ew_rows(long long N, long long D, const float* __restrict__ params,
                                                                   const float* __restrict__ q, float* __restrict__ logp,
                                                                   float* __restrict__ grad) {{
  const int lane = threadIdx.x & 63;
  const int waves = blockDim.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * waves + (threadIdx.x >> 6); r < N; r += (int64_t)gridDim.x * waves) {{
    const float* qr = q + r * D;
    float* gr = grad + r * D;
    {acc_decl}
    if (D % 4 == 0) {{
      F4 xq[4]; {p_decl_arr}
      row_sweep4<4>(lane, D,
        [&](int u, int64_t j) {{ xq[u] = ld4(qr + j); {p_load_u}}},
        [&](int u, int64_t j) {{
          const float xs[4] = {{xq[u].x, xq[u].y, xq[u].z, xq[u].w}}; {p_unpack_u}
          float gs[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {{
            const float x = xs[e]; {p_decl}
            {body}
            gs[e] = {g_expr};
            {acc_add_rows}
          }}
          st4(gr + j, F4{{gs[0], gs[1], gs[2], gs[3]}});
        }});
    }} else {{
      for (int64_t j = lane; j < D; j += 64) {{
        const float x = qr[j]; {p_scalar}
        {body}
        gr[j] = {g_expr};
        {acc_add_rows}
      }}
    }}
    const float lp = (float)({lp_expr});
    if (lane == 0) logp[r] = lp;
  }}
}}
"""
    params = torch.stack(gen.params).contiguous() if gen.params else None
    src = ElementwiseSource(source, params, len(used),
                            f"{len(gen.lines)} fp32 ops per element, {n_p} parameter vector(s), {len(used)} row sum(s)")
    src.rows_source = rows_source
    return src


class _Wrap(torch.nn.Module):
    """``torch.fx`` traces modules; tensors the function closes over become ``get_attr`` constants."""

    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def forward(self, q):
        return self._fn(q)


ROWS_TU = """#include "bjx_traj_dev.h"
using namespace bjx;
%(source)s
"""


def from_elementwise(fn: Callable, dim: int, device="cuda"):
    """A ``blackjax_amd.targets.DeviceTarget`` computing ``(fn(q), d fn / d q)`` in one launch (see the module
    docstring).  ``dim`` = D.  Rows of at most 1 024 floats with ``D % 4 == 0`` give a full ``DeviceTarget`` (usable
    with ``fuse_target=True`` too); any other D gives an ``ElementwiseRowsTarget``: the same generated arithmetic in a
    row-loop kernel, an external callable only (BASELINE.json configs[3], D = 4 096, is served by it)."""
    from .targets import DeviceTarget, ElementwiseRowsTarget

    dim = int(dim)
    if dim < 1:
        raise ValueError("from_elementwise: dim must be >= 1")
    src = trace(fn, dim, device)
    if dim % 4 != 0 or dim > 1024:
        return ElementwiseRowsTarget(src, dim)
    tgt = DeviceTarget(src.source, src.params)
    tgt.elementwise = src
    return tgt
