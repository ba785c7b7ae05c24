"""Host-side plumbing shared by the kernels' Python drivers."""
from __future__ import annotations

import threading
import weakref
from typing import Callable

import torch

_VG_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_CAPTURE_LOCK = threading.RLock()
_CAPTURE_TLS = threading.local()  # .depth > 0: this thread is inside a stream capture
_GRAVEYARD: list = []  # graphs whose last reference died inside a capture of the SAME thread: destroyed right after it


class LockedGraph(torch.cuda.CUDAGraph):
    """``torch.cuda.CUDAGraph`` whose DESTRUCTION is serialised with stream captures.  On ROCm 7 destroying a graph
    (``hipGraphExecDestroy`` / ``hipGraphDestroy`` / the release of its private pool) in one thread while another thread
    is between ``capture_begin`` and ``capture_end`` aborts the process -- seen as ``Fatal Python error: Aborted`` in 3 of
    9 runs of the GPU suite (round 6: one thread dropping a NUTS workspace, or finishing and releasing its algorithm
    objects, while the other recorded a tail sequence; ``gpurun_out/soak`` logs, NOTEBOOK section 19.2).  Every graph the
    drivers record is one of these: its last reference going away -- by reference count or by the cyclic collector, in
    whatever thread -- waits for the capture lock before the graph is reset; inside a capture of the same thread the
    reset is deferred to the end of that capture."""

    def __del__(self):
        try:
            if getattr(_CAPTURE_TLS, "depth", 0) > 0:
                _GRAVEYARD.append(self)  # (resurrected: reset explicitly by _drain_graveyard)
                return
            with _CAPTURE_LOCK:
                self.reset()
        except Exception:  # interpreter shutdown, a graph that never captured anything
            pass


def new_graph() -> "LockedGraph":
    return LockedGraph()


def _drain_graveyard():
    while _GRAVEYARD:
        g = _GRAVEYARD.pop()
        try:
            g.reset()
        except Exception:
            pass


def record_graph(graph, **kw):
    """``torch.cuda.graph(graph, capture_error_mode="thread_local")`` under a process-wide lock.  Two stream captures
    under way at once -- two Python threads stepping two algorithm objects, each recording its inner loop -- crash the
    process on ROCm 7 (SIGSEGV / abort inside the capture, measured round 6: tools/scratch/thread_probe.py), and so does
    a graph destroyed by one thread while another captures (``LockedGraph``), so recordings AND graph destruction are
    serialised; replays, plain launches and allocations of other threads are not."""
    import contextlib
    import sys

    lock = _CAPTURE_LOCK

    @contextlib.contextmanager
    def ctx():
        with lock:
            cm = torch.cuda.graph(graph, capture_error_mode="thread_local", **kw)  # other threads (the RCCL watchdog) may poll events
            # its __enter__ runs gc.collect() BEFORE capture_begin: dead workspaces' graphs are reset there (depth still 0,
            # the lock is ours), not inside the capture
            cm.__enter__()
            _CAPTURE_TLS.depth = getattr(_CAPTURE_TLS, "depth", 0) + 1
            ok = False
            try:
                yield
                ok = True
            finally:
                try:
                    cm.__exit__(None, None, None) if ok else cm.__exit__(*sys.exc_info())
                finally:
                    _CAPTURE_TLS.depth -= 1
                    if _CAPTURE_TLS.depth == 0:
                        _drain_graveyard()

    return ctx()


def value_and_grad(logdensity_fn: Callable) -> Callable:
    """Batched counterpart of ``jax.value_and_grad(logdensity_fn)`` (blackjax/mcmc/hmc.py:91,
    integrators.py:189): returns ``f(q: (N, D)) -> (logp: (N,), grad: (N, D))``.

    ``logdensity_fn`` is a PyTorch callable over the whole batch.  It may either
    return ``(logp, grad)`` itself (fast path, e.g. ``blackjax_amd.targets``) or
    return only ``logp`` with an autograd graph, in which case the gradient is taken
    with ``torch.autograd.grad`` (chains are independent, so d sum(logp) / dq is the
    per-chain gradient).
    """
    if getattr(logdensity_fn, "_bjx_value_and_grad", False):
        return logdensity_fn
    try:
        cached = _VG_CACHE.get(logdensity_fn)
    except TypeError:
        cached = None
    if cached is not None:
        return cached

    mode = {"kind": None}

    if getattr(logdensity_fn, "_bjx_returns_pair", False):
        mode["kind"] = "pair"  # declared with blackjax_amd.returns_pair: never probed under autograd

    def _grad(lp, q):
        # chains are independent, so the vector-Jacobian product with ones IS the per-chain gradient:
        # grad_outputs = an expanded 0-d one (no fill kernel, no sum kernel, no backward-of-sum kernel)
        ones = _one(lp).expand_as(lp)
        (g,) = torch.autograd.grad(lp, q, grad_outputs=ones)
        return g

    def _autograd(q):
        q = q.detach().requires_grad_(True)
        with torch.enable_grad():
            lp = logdensity_fn(q)
            g = _grad(lp, q)
        return lp.detach(), g

    ew: dict = {}  # (D, device) -> generated element-wise target, or None (not of that shape / failed its check)

    calls: dict = {}  # (D, device) -> calls served by the generated kernel

    def _autograd_or_elementwise(q, first=None):
        key = (int(q.shape[-1]), q.device)
        if key in ew:
            tgt = ew[key]
            if tgt is None:
                return first if first is not None else _autograd(q)
            n = calls[key] = calls.get(key, 0) + 1
            if n in _RECHECK_AT or n % _RECHECK_EVERY == 0:
                # A traced function is the function AS IT WAS at its first call (as under jax.jit).  A closure over Python
                # state that changes later -- a tempering beta held as a float, a minibatch index -- would go stale
                # silently, so the generated kernel is re-checked against eager autograd on the live batch at calls 16,
                # 256 and every 4 096th (one autograd pass each); a disagreement puts the callable back on autograd.
                if not (q.is_cuda and torch.cuda.is_current_stream_capturing()):
                    lp_a, g_a = _autograd(q)
                    lp_k, g_k = tgt(q)
                    if _close(lp_k, lp_a) and _close(g_k, g_a):
                        return lp_k, g_k
                    ew[key] = None
                    import warnings

                    warnings.warn(
                        "blackjax_amd: the kernel generated from this log-density no longer agrees with the function "
                        f"itself (call {n}: it reads Python state that changed after its first call?) -- evaluating it "
                        "eagerly under torch.autograd from here on; declare such callables with blackjax_amd.no_trace.",
                        RuntimeWarning, stacklevel=3)
                    return lp_a, g_a
            return tgt(q)
        lp, g = first if first is not None else _autograd(q)
        ew[key] = _try_elementwise(logdensity_fn, q, lp, g)
        return lp, g

    def vg(q):
        if mode["kind"] == "autograd":
            return _autograd_or_elementwise(q)
        if mode["kind"] == "pair":
            return logdensity_fn(q)
        # First call of an UNDECLARED callable: one evaluation decides which kind it is.  It is made
        # under autograd (so a callable that returns only logp needs no second pass); a callable that
        # returns (logp, grad) itself should be declared with ``blackjax_amd.returns_pair`` -- it is
        # then never run on a requires_grad leaf (no throw-away autograd graph over the (N, D) batch,
        # in-place updates of q and host reads inside the callable keep working).
        qg = q.detach().requires_grad_(True)
        with torch.enable_grad():
            out = logdensity_fn(qg)
            if isinstance(out, (tuple, list)) and len(out) == 2:
                mode["kind"] = "pair"
                lp, g = out[0].detach(), out[1].detach()
                del out
                return lp, g
            mode["kind"] = "autograd"
            g = _grad(out, qg)
        # a log-density that returns only logp: what jax.value_and_grad + XLA fuse in the reference
        # (integrators.py:189,204).  Try the one-kernel form first (``elementwise.from_elementwise``: torch.fx trace ->
        # forward-mode derivative -> ONE generated HIP value-and-gradient kernel), checked here against this very
        # autograd evaluation; anything it cannot express, or that fails the check, stays on autograd.
        return _autograd_or_elementwise(q, first=(out.detach(), g))

    vg._bjx_elementwise = ew
    vg._bjx_value_and_grad = True
    try:
        _VG_CACHE[logdensity_fn] = vg
    except TypeError:
        pass
    return vg


_WARNED_AUTOGRAD: set = set()
_RECHECK_AT = (16, 256)   # calls at which a generated kernel is re-checked against eager autograd ...
_RECHECK_EVERY = 4096     # ... and every this many calls after


def _close(a, b) -> bool:
    """Generated kernel vs autograd on one batch: within 1e-4 of the batch's largest finite magnitude, same non-finite
    entries (rounding differs by ~1e-6; a mis-traced or stale function by far more)."""
    a, b = a.float(), b.float()
    fin = torch.isfinite(b)
    scale = float(b[fin].abs().max()) if bool(fin.any()) else 1.0
    same_nonfinite = bool(((a == b) | (torch.isnan(a) & torch.isnan(b)))[~fin].all())
    return same_nonfinite and bool(((a - b)[fin].abs() <= 1e-4 * max(scale, 1e-30)).all())


def no_trace(logdensity_fn: Callable) -> Callable:
    """Declare that a log-density must NOT be traced into a generated kernel (it reads Python state that changes
    between calls, or its eager arithmetic is wanted as is): it is always evaluated eagerly under autograd."""
    try:
        logdensity_fn._bjx_no_trace = True
        return logdensity_fn
    except AttributeError:
        w = _Capturable(logdensity_fn)
        w._bjx_capturable = False
        w._bjx_no_trace = True
        return w


def _try_elementwise(logdensity_fn, q, lp_ref, g_ref):
    """The default fast path for a plain PyTorch log-density: ``targets.from_elementwise`` on the
    user's function, accepted only if its (logp, grad) at ``q`` agree with the autograd evaluation just made
    (1e-4 of the batch's largest magnitude: a mis-traced function -- data-dependent Python control flow that torch.fx
    froze, state read at trace time -- differs by far more; rounding differs by ~1e-6).  Returns the target or None;
    never raises.  Like ``jax.jit`` in the reference, a traced function is the function AS IT WAS when first called:
    ``blackjax_amd.no_trace(fn)`` (or ``BJX_AUTO_ELEMENTWISE=0``) keeps a callable on eager autograd."""
    import os
    import warnings

    why = None
    if os.environ.get("BJX_AUTO_ELEMENTWISE", "1") == "0" or getattr(logdensity_fn, "_bjx_no_trace", False):
        why = "tracing disabled for this callable"
    elif not (q.is_cuda and q.ndim == 2 and q.dtype == torch.float32 and q.shape[0] > 0):
        why = "not a float32 (N, D) device batch"
    elif torch.cuda.is_current_stream_capturing():
        why = "first call made under stream capture"
    tgt = None
    if why is None:
        try:
            from .targets import from_elementwise

            tgt = from_elementwise(logdensity_fn, int(q.shape[1]), device=q.device)
            lp, g = tgt(q.detach().contiguous())

            if not (_close(lp, lp_ref) and _close(g, g_ref)):
                tgt, why = None, "the generated kernel disagreed with autograd on the first batch"
        except NotImplementedError as e:
            tgt, why = None, str(e)
        except Exception as e:  # hiprtc / tracing trouble never takes the sampler down: autograd serves the call
            tgt, why = None, f"{type(e).__name__}: {e}"
    if tgt is None and id(logdensity_fn) not in _WARNED_AUTOGRAD and "disabled" not in (why or ""):
        _WARNED_AUTOGRAD.add(id(logdensity_fn))
        warnings.warn(
            "blackjax_amd: this log-density is evaluated eagerly under torch.autograd (about a dozen element-wise passes "
            f"per gradient: ~3 x slower than the sampler's kernels at large N x D) -- {why}.  Return (logp, grad) "
            "yourself (blackjax_amd.returns_pair), use blackjax_amd.targets, or torch.compile the pair.",
            RuntimeWarning, stacklevel=4)
    return tgt


_ONES: dict = {}


def _one(like: torch.Tensor) -> torch.Tensor:
    key = (like.device, like.dtype)
    t = _ONES.get(key)
    if t is None:
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            # a tensor created under capture lives in the graph's private pool and its fill is only
            # recorded, never run: do not cache it -- the recorded graph owns this one
            return torch.ones((), device=like.device, dtype=like.dtype)
        t = _ONES[key] = torch.ones((), device=like.device, dtype=like.dtype)
    return t


def returns_pair(logdensity_fn: Callable) -> Callable:
    """Declare that ``logdensity_fn(q)`` returns ``(logp, grad)`` itself (hand-written gradient, a
    fused kernel, ``torch.func`` ...).  Undeclared callables are classified by their first call, which
    runs under autograd on a ``requires_grad`` leaf; a declared one is never probed."""
    try:
        logdensity_fn._bjx_returns_pair = True
        return logdensity_fn
    except AttributeError:
        w = _Capturable(logdensity_fn)
        w._bjx_capturable = False
        w._bjx_returns_pair = True
        return w


_WARNED_EAGER: set = set()


def warn_eager_driver(logdensity_fn, what: str) -> None:
    """One-time note that an undeclared callable is driven with plain launches (host-bound for
    short launches); ``blackjax_amd.capturable`` or ``use_graph=True`` selects the recorded driver."""
    k = id(logdensity_fn)
    if k in _WARNED_EAGER:
        return
    _WARNED_EAGER.add(k)
    import warnings

    warnings.warn(
        f"blackjax_amd.{what}: this log-density callable is not declared recordable, so it is driven "
        "with plain launches (several Python launches per leapfrog, host-bound when launches are short). "
        "Wrap it in blackjax_amd.capturable(fn) if it has static shapes, does not synchronise with the "
        "host and reads no Python state that changes between calls, or pass use_graph=True.",
        RuntimeWarning, stacklevel=3)


class _Capturable:
    """Wrapper for callables that cannot carry attributes (bound methods, builtins)."""

    _bjx_capturable = True

    def __init__(self, fn):
        self._fn = fn

    def __call__(self, q):
        return self._fn(q)


def capturable(logdensity_fn: Callable) -> Callable:
    """Declare a log-density callable safe to RECORD in a HIP graph and replay (what ``jax.jit`` does
    to the reference's ``logdensity_fn``): static shapes, no host synchronisation, and no Python-side
    state that changes between calls (a minibatch index, a tempering ``beta`` held as a float ...),
    because a replay re-runs the recorded kernels, not the Python code.  The default drivers
    (``use_graph="auto"``) record only callables declared this way -- ``blackjax_amd.targets`` are --
    and drive every other callable with plain launches; ``use_graph=True`` records regardless."""
    try:
        logdensity_fn._bjx_capturable = True
        return logdensity_fn
    except AttributeError:
        return _Capturable(logdensity_fn)


def is_capturable(logdensity_fn) -> bool:
    return bool(getattr(logdensity_fn, "_bjx_capturable", False))


def check_batch(x: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(x)}")
    if not x.is_cuda:
        raise RuntimeError(
            f"{name} lives on {x.device}: blackjax_amd runs on ROCm device tensors only "
            "(there is no CPU fallback)"
        )
    if x.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {x.dtype}")
    return x.contiguous()


def eval_logdensity(vg: Callable, q: torch.Tensor):
    """Call the user's value-and-grad and normalise its outputs to contiguous fp32."""
    logp, g = vg(q)
    if logp.dtype != torch.float32:
        logp = logp.float()
    if g.dtype != torch.float32:
        g = g.float()
    if g.shape != q.shape or logp.shape != q.shape[:1]:
        raise ValueError(
            f"logdensity_fn must return logp of shape {tuple(q.shape[:1])} and grad of shape "
            f"{tuple(q.shape)}; got {tuple(logp.shape)} and {tuple(g.shape)}"
        )
    return logp.contiguous(), g.contiguous()


def step_size_args(step_size, n_chains: int, device):
    """-> (scalar eps, per-chain tensor or None)."""
    if isinstance(step_size, torch.Tensor):
        if step_size.ndim == 0 and not step_size.is_cuda:
            return float(step_size), None
        t = step_size.to(device=device, dtype=torch.float32)
        if t.ndim == 0:
            t = t.expand(n_chains)
        if t.shape != (n_chains,):
            raise ValueError(f"per-chain step_size must have shape ({n_chains},), got {tuple(t.shape)}")
        return 0.0, t.contiguous()
    return float(step_size), None
