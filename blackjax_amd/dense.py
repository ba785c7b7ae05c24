"""Launch helpers for the dense Gaussian-Euclidean metric (``bjx_dense.hip``).

``metric.kind == "dense"``: one (D, D) matrix shared by all chains -> fp32 MFMA GEMMs.
``metric.kind == "dense_pc"``: one (D, D) matrix per chain, (N, D, D) -> batched fp64-accumulated
matrix-vector kernels (``*_dense_pc`` entry points with ``matrix_stride = D*D``).
"""
from __future__ import annotations

import torch

from . import _lib


def _imm_ptr(metric, n: int, d: int) -> int:
    """Pointer to hand to the shared-matrix GEMM entry points as ``imm``.  ``v = imm p`` reads the matrix AS STORED
    (``v[i] = sum_k imm[i][k] p[k]``, blackjax/mcmc/metrics.py:263-304); whole 128 x 128 tiles run on the kernel that
    does exactly that, ragged shapes on the general kernel, which walks ``B[k][n]`` -- so it gets the transposed copy
    and a matrix that is symmetric only up to rounding (a Welford covariance) gives a chain the same product whatever
    the size of the batch it runs in (round 4)."""
    if metric.kind == "dense" and metric.imm_t is not None and not (n % 128 == 0 and d % 128 == 0):
        return metric.imm_t.data_ptr()
    return metric.imm.data_ptr()


def momentum(stream, metric, k0, k1, off, fold, n, d, p0, ke0, force_pc: bool = False):
    """``force_pc``: route a shared matrix through the fp64-accumulated kernels too (NUTS)."""
    z = torch.empty_like(p0)
    v = torch.empty_like(p0)
    if metric.kind == "dense_pc" or force_pc:
        stride = d * d if metric.kind == "dense_pc" else 0
        _lib.call("bjx_hmc_momentum_dense_pc", stream, k0, k1, off, fold, n, d,
                  metric.mass_sqrt_t.data_ptr(), metric.imm.data_ptr(), stride, z.data_ptr(),
                  v.data_ptr(), p0.data_ptr(), ke0.data_ptr())
    else:
        _lib.call("bjx_hmc_momentum_dense", stream, k0, k1, off, fold, n, d,
                  metric.mass_sqrt_t.data_ptr(), _imm_ptr(metric, n, d), z.data_ptr(), v.data_ptr(),
                  p0.data_ptr(), ke0.data_ptr())
    return v  # imm @ p0 (velocity of the initial state)


def leapfrog(stream, metric, n, d, n_kicks, eps, eps_pc, q_in, p_in, g, q_out, p_out):
    """Returns the tensor that holds the new momentum (the shared-matrix GEMM cannot update p in
    place: its column blocks re-read the un-kicked momentum)."""
    if metric.kind == "dense_pc":
        _lib.call("bjx_leapfrog_dense_pc", stream, n, d, n_kicks, eps, _lib.ptr(eps_pc),
                  metric.imm.data_ptr(), d * d, q_in.data_ptr(), p_in.data_ptr(), g.data_ptr(),
                  q_out.data_ptr(), p_out.data_ptr())
        return p_out
    if p_out.data_ptr() == p_in.data_ptr():
        p_out = torch.empty_like(p_in)
    _lib.call("bjx_leapfrog_dense", stream, n, d, n_kicks, eps, _lib.ptr(eps_pc),
              _imm_ptr(metric, n, d), q_in.data_ptr(), p_in.data_ptr(), g.data_ptr(),
              q_out.data_ptr(), p_out.data_ptr())
    return p_out


def finish(stream, metric, k0, k1, off, fold, n, d, eps, eps_pc, thr, q0, logp0, g0, ke0, q, logp, g,
           p, p_end, q_new, logp_new, g_new, acc_rate, is_acc, is_div, energy):
    p1 = torch.empty_like(p)
    v = torch.empty_like(p)
    head = [stream, k0, k1, off, fold, n, d, eps, _lib.ptr(eps_pc), _imm_ptr(metric, n, d)]
    tail = [thr, q0.data_ptr(), logp0.data_ptr(), g0.data_ptr(), ke0.data_ptr(), q.data_ptr(),
            logp.data_ptr(), g.data_ptr(), p.data_ptr(), p1.data_ptr(), v.data_ptr(),
            p_end.data_ptr(), q_new.data_ptr(), logp_new.data_ptr(), g_new.data_ptr(),
            acc_rate.data_ptr(), is_acc.data_ptr(), is_div.data_ptr(), energy.data_ptr()]
    if metric.kind == "dense_pc":
        _lib.call("bjx_hmc_finish_dense_pc", *head, d * d, *tail)
    else:
        _lib.call("bjx_hmc_finish_dense", *head, *tail)


def matrix_stride(metric, d: int) -> int:
    """``matrix_stride`` argument of the ``*_dense_coef`` entry points: < 0 = one shared matrix on
    the MFMA GEMM path, ``d * d`` = one matrix per chain on the fp64 matrix-vector path."""
    return d * d if metric.kind == "dense_pc" else -1


def leapfrog_coef(stream, metric, n, d, n_kicks, kick_a, kick_b, drift, eps, eps_pc, q_in, p_in, g,
                  q_out, p_out, n_steps=None, step_idx: int = 0):
    """One position update of a general palindromic integrator (kicks ``eps*kick_a`` [, ``eps*kick_b``],
    drift ``eps*drift``), optionally masked by per-chain trajectory lengths.  Returns the tensor that
    holds the new momentum (the shared-matrix GEMM cannot update p in place)."""
    ms = matrix_stride(metric, d)
    if ms < 0 and p_out.data_ptr() == p_in.data_ptr():
        p_out = torch.empty_like(p_in)
    _lib.call("bjx_leapfrog_dense_coef", stream, n, d, n_kicks, kick_a, kick_b, drift, eps, _lib.ptr(eps_pc),
              _imm_ptr(metric, n, d), ms, q_in.data_ptr(), p_in.data_ptr(), g.data_ptr(), q_out.data_ptr(),
              p_out.data_ptr(), _lib.ptr(n_steps), int(step_idx))
    return p_out


def finish_coef(stream, metric, k0, k1, off, fold, n, d, kick_coef, eps, eps_pc, thr, q0, logp0, g0, ke0, q,
                logp, g, p, p_end, q_new, logp_new, g_new, acc_rate, is_acc, is_div, energy):
    p1 = torch.empty_like(p)
    v = torch.empty_like(p)
    _lib.call("bjx_hmc_finish_dense_coef", stream, k0, k1, off, fold, n, d, kick_coef, eps, _lib.ptr(eps_pc),
              _imm_ptr(metric, n, d), matrix_stride(metric, d), thr, q0.data_ptr(), logp0.data_ptr(),
              g0.data_ptr(), ke0.data_ptr(), q.data_ptr(), logp.data_ptr(), g.data_ptr(), p.data_ptr(),
              p1.data_ptr(), v.data_ptr(), p_end.data_ptr(), q_new.data_ptr(), logp_new.data_ptr(),
              g_new.data_ptr(), acc_rate.data_ptr(), is_acc.data_ptr(), is_div.data_ptr(), energy.data_ptr())


def mhmc_step(stream, metric, k0, k1, off, fold, n, d, step, eps, eps_pc, thr, logp0, ke0, q, p, g, logp,
              weight, slpa, any_div, ever, pq, pp, pg, plogp, penergy, n_steps=None, kick_coef=None):
    """Closing half kick + reservoir step of multinomial HMC; returns the fully kicked momentum.
    ``n_steps``: per-chain trajectory lengths (dmhmc) -- chains with ``step >= n_steps`` are left alone.
    ``kick_coef``: closing-kick coefficient b1 of a general palindromic integrator (None: velocity Verlet)."""
    p1 = torch.empty_like(p)
    v = torch.empty_like(p)
    args = (stream, k0, k1, off, fold, n, d, step, eps, _lib.ptr(eps_pc),
            _imm_ptr(metric, n, d), matrix_stride(metric, d), thr, logp0.data_ptr(), ke0.data_ptr(),
            q.data_ptr(), p.data_ptr(), g.data_ptr(), logp.data_ptr(), p1.data_ptr(), v.data_ptr(),
            weight.data_ptr(), slpa.data_ptr(), any_div.data_ptr(), ever.data_ptr(), pq.data_ptr(),
            pp.data_ptr(), pg.data_ptr(), plogp.data_ptr(), penergy.data_ptr())
    if kick_coef is not None:  # any palindromic integrator: closing kick (eps * b1) g (round 4)
        _lib.call("bjx_mhmc_step_dense_coef", *args[:8], float(kick_coef), *args[8:], _lib.ptr(n_steps))
    elif n_steps is None:
        _lib.call("bjx_mhmc_step_dense", *args)
    else:
        _lib.call("bjx_mhmc_step_dense_masked", *args, n_steps.data_ptr())
    return p1
