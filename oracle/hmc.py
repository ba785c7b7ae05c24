"""Oracle for the HMC transition (TEST INFRASTRUCTURE, see package docstring).

Batched restatement: every array carries a leading chain axis N; chain ``i``
of ``kernel(rng_key, state, ...)`` equals the reference's single-chain
``blackjax.hmc.build_kernel()(jax.random.split(rng_key, N)[i], state_i, ...)``
i.e. the "step-major" vmap layout of
docs/examples/howto_sample_multiple_chains.md:120-127.

Reference lines followed
* HMCState / HMCInfo / init          blackjax/mcmc/hmc.py:38-92
* velocity_verlet one_step           blackjax/mcmc/integrators.py:104-150,191-205,226-243,321-322
* static_integration                 blackjax/mcmc/trajectory.py:136-167
* gaussian_euclidean metric          blackjax/mcmc/metrics.py:221-304,701-729
* generate_gaussian_noise/linear_map blackjax/util.py:23-61,66-91
* hmc_energy                         blackjax/mcmc/trajectory.py:730-750
* safe_energy_diff / static_binomial blackjax/mcmc/proposal.py:45-48,214-235
* hmc_proposal.generate / kernel     blackjax/mcmc/hmc.py:153-176,279-312
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np

from . import prng
from .fp import dot64, exp_cr, f32, f64, fma32


class HMCState(NamedTuple):  # hmc.py:38-49
    position: np.ndarray  # (N, D)
    logdensity: np.ndarray  # (N,)
    logdensity_grad: np.ndarray  # (N, D)


class IntegratorState(NamedTuple):  # integrators.py:43-53
    position: np.ndarray
    momentum: np.ndarray
    logdensity: np.ndarray
    logdensity_grad: np.ndarray


class HMCInfo(NamedTuple):  # hmc.py:52-87
    momentum: np.ndarray
    acceptance_rate: np.ndarray
    is_accepted: np.ndarray
    is_divergent: np.ndarray
    energy: np.ndarray
    proposal: IntegratorState
    num_integration_steps: int


def init(position, logdensity_fn: Callable) -> HMCState:  # hmc.py:90-92
    position = np.asarray(position, dtype=f32)
    logp, grad = logdensity_fn(position)
    return HMCState(position, np.asarray(logp, f32), np.asarray(grad, f32))


# ----------------------------------------------------------------------------- metric
class Metric(NamedTuple):
    inverse_mass_matrix: np.ndarray  # (D,), (N, D) or (D, D)
    mass_matrix_sqrt: np.ndarray  # diag: 1/sqrt(imm) ; dense: L^{-T}
    is_dense: bool
    # how a SHARED dense matrix is applied: "f64" = fp64-accumulated dot rounded once (order
    # independent); "f32chain" = fp32 fmaf chain in the k order of the engine's MFMA GEMMs
    # (fp.mfma_k_order) -- the arithmetic of an fp32 "precision=highest" dot for that order
    dense_accum: str = "f64"


def default_metric(inverse_mass_matrix, n_chains=None, per_chain_diag=False,
                   dense_accum="f64", mass_matrix_sqrt=None) -> Metric:
    """metrics.py:180-218 -> gaussian_euclidean 221-346 -> _format_covariance 701-729.

    diag: ``inv_cov_sqrt = sqrt(imm)``, ``mass_matrix_sqrt = 1/inv_cov_sqrt``
    (704-709).  A 2-d array whose leading dimension equals ``n_chains`` and is not
    square is a per-chain diagonal (the vmapped-warmup case).  dense:
    ``L = cholesky(imm, lower)``, ``mass_matrix_sqrt = solve_triangular(L, I,
    lower=True, trans=True) = L^{-T}`` (711-715); factorised in fp64 and rounded
    once to fp32.  ``mass_matrix_sqrt`` (dense only) overrides the factor: two fp64 LAPACK
    builds may round a handful of the D^2 entries of L^{-T} differently, so a bit-exact
    comparison hands both sides the SAME fp32 factor (checked to be within 1 ulp of this one).
    """
    imm = np.asarray(inverse_mass_matrix, dtype=f32)
    if imm.ndim == 1 or (imm.ndim == 2 and per_chain_diag) or (
            imm.ndim == 2 and n_chains is not None and imm.shape[0] == n_chains
            and imm.shape[0] != imm.shape[1]):
        inv_sqrt = np.sqrt(imm)
        return Metric(imm, (f32(1.0) / inv_sqrt).astype(f32), False)
    if imm.ndim == 2 and imm.shape[0] == imm.shape[1]:
        if mass_matrix_sqrt is None:
            L = np.linalg.cholesky(imm.astype(f64))
            mass_matrix_sqrt = np.linalg.solve(L.T, np.eye(L.shape[0]))  # L^{-T}
        return Metric(imm, np.asarray(mass_matrix_sqrt).astype(f32), True, dense_accum)
    if imm.ndim == 3 and imm.shape[1] == imm.shape[2]:
        # one dense matrix PER CHAIN (what a vmapped dense window_adaptation produces)
        if mass_matrix_sqrt is None:
            L = np.linalg.cholesky(imm.astype(f64))
            eye = np.broadcast_to(np.eye(imm.shape[1]), imm.shape)
            mass_matrix_sqrt = np.linalg.solve(np.swapaxes(L, 1, 2), eye)  # L^{-T} per chain
        return Metric(imm, np.asarray(mass_matrix_sqrt).astype(f32), True)
    raise ValueError(
        "The mass matrix has the wrong number of dimensions:"
        f" expected 1 or 2, got {imm.ndim}."
    )


def linear_map(metric: Metric, mat, x):
    """util.py:23-61: diag -> elementwise multiply; dense -> mat @ x per chain
    (fp64 accumulate, rounded once)."""
    if not metric.is_dense:
        return (mat * x).astype(f32)
    if mat.ndim == 3:  # per-chain matrices
        return np.einsum("nij,nj->ni", mat.astype(f64), x.astype(f64)).astype(f32)
    if metric.dense_accum == "f32chain":  # y[n] = fmaf chain over k of x[k] * mat[n][k]
        from . import cport
        from .fp import mfma_k_order

        return cport.gemm_f32chain(x, np.ascontiguousarray(mat, dtype=f32).T, mfma_k_order(mat.shape[1]))
    return (x.astype(f64) @ mat.astype(f64).T).astype(f32)


def sample_momentum(metric: Metric, keys, D):
    """metrics.py:260-261 -> util.py:89-91: p = mass_matrix_sqrt (.) normal(key, (D,))."""
    z = prng.normal(keys, (D,))
    return linear_map(metric, metric.mass_matrix_sqrt, z)


def kinetic_energy(metric: Metric, p):
    """metrics.py:263-270: 0.5 * dot(linear_map(imm, p), p)."""
    v = linear_map(metric, metric.inverse_mass_matrix, p)
    return f32(0.5) * dot64(v, p)


# ----------------------------------------------------------------------------- integrator
def _col(x):
    x = np.asarray(x, dtype=f32)
    return x[:, None] if x.ndim == 1 else x


def velocity_verlet(state: IntegratorState, step_size, logdensity_fn, metric: Metric):
    """One velocity-Verlet step, coefficients [0.5, 1.0, 0.5]
    (integrators.py:321-322 -> 104-150).  ``step_size`` scalar or (N,) (may be
    negative: direction*step_size in NUTS, trajectory.py:323).

    p += (eps*0.5)*g ; v = dK/dp = imm p ; q += (eps*1.0)*v ; (logp,g)=f(q) ; p += (eps*0.5)*g
    each ``x + s*y`` a single fma (see package docstring).
    """
    q, p, _, g = state
    eps = _col(step_size) if np.ndim(step_size) else f32(step_size)
    h = (eps * f32(0.5)).astype(f32) if np.ndim(eps) else f32(eps * f32(0.5))
    p = fma32(h, g, p)
    v = linear_map(metric, metric.inverse_mass_matrix, p)
    q = fma32(eps, v, q)
    logp, g = logdensity_fn(q)
    p = fma32(h, g, p)
    return IntegratorState(q, p, np.asarray(logp, f32), np.asarray(g, f32))


def hmc_energy(metric: Metric, state: IntegratorState):
    """trajectory.py:745-748: -logdensity + kinetic_energy(momentum)."""
    return (-state.logdensity + kinetic_energy(metric, state.momentum)).astype(f32)


def safe_energy_diff(e0, e1):
    """proposal.py:45-48."""
    with np.errstate(invalid="ignore"):
        d = (np.asarray(e0, f32) - np.asarray(e1, f32)).astype(f32)
    return np.where(np.isnan(d), f32(-np.inf), d).astype(f32)


# ----------------------------------------------------------------------------- kernel
def chain_keys(rng_key, n, chain_offset=0):
    """split(rng_key, N_total)[offset:offset+n] -- per-chain keys (step-major layout)."""
    return prng.split(rng_key, n, offset=chain_offset)


def integrator_step(z: IntegratorState, step_size, logdensity_fn, metric: Metric, coefficients=None):
    """One step of the palindromic integrator ``coefficients`` ([b1, a1, ..., b1], integrators.py:104-150;
    ``None`` = velocity Verlet)."""
    if coefficients is None:
        return velocity_verlet(z, step_size, logdensity_fn, metric)
    from . import integrators as _oi  # (imports this module)

    return _oi.one_step(z, step_size, logdensity_fn, metric, coefficients)


def kernel(rng_key, state: HMCState, logdensity_fn, step_size, inverse_mass_matrix,
           num_integration_steps: int, divergence_threshold: float = 1000.0,
           chain_offset: int = 0, chain_keys_override=None, per_chain_diag=False, metric=None,
           coefficients=None):
    """hmc.py:279-312 with hmc_proposal.generate 153-176, batched over chains.  ``metric``: a
    prepared ``Metric`` (e.g. ``default_metric(..., dense_accum="f32chain")``) instead of
    classifying ``inverse_mass_matrix`` again."""
    N, D = state.position.shape
    if metric is None:
        metric = default_metric(inverse_mass_matrix, n_chains=N, per_chain_diag=per_chain_diag)
    keys = chain_keys(rng_key, N, chain_offset) if chain_keys_override is None else chain_keys_override
    kk = prng.split(keys, 2)  # hmc.py:299
    key_momentum, key_integrator = kk[:, 0], kk[:, 1]

    p0 = sample_momentum(metric, key_momentum, D)  # hmc.py:302
    z0 = IntegratorState(state.position, p0, state.logdensity, state.logdensity_grad)
    z = z0
    for _ in range(num_integration_steps):  # trajectory.py:155-165
        z = integrator_step(z, step_size, logdensity_fn, metric, coefficients)
    end = IntegratorState(z.position, (f32(-1.0) * z.momentum).astype(f32), z.logdensity,
                          z.logdensity_grad)  # flip_momentum hmc.py:95-112
    e0 = hmc_energy(metric, z0)
    e1 = hmc_energy(metric, end)
    delta = safe_energy_diff(e0, e1)
    is_div = (-delta) > f32(divergence_threshold)  # hmc.py:162
    p_acc = np.minimum(exp_cr(delta), f32(1.0))  # proposal.py:225
    u = prng.uniform(key_integrator, ())  # bernoulli, proposal.py:226
    acc = u < p_acc
    new_state = HMCState(
        np.where(acc[:, None], end.position, state.position).astype(f32),
        np.where(acc, end.logdensity, state.logdensity).astype(f32),
        np.where(acc[:, None], end.logdensity_grad, state.logdensity_grad).astype(f32),
    )
    info = HMCInfo(p0, p_acc, acc, is_div, e1, end, num_integration_steps)
    return new_state, info


def mhmc_kernel(rng_key, state: HMCState, logdensity_fn, step_size, inverse_mass_matrix,
                num_integration_steps: int, divergence_threshold: float = 1000.0,
                chain_offset: int = 0, chain_keys_override=None, per_chain_diag=False, metric=None,
                coefficients=None):
    """blackjax.mhmc: hmc.build_kernel(build_proposal=multinomial_hmc_proposal)
    (hmc.py:181-248, 279-312) with static_progressive_integration (trajectory.py:170-232) and
    progressive_uniform_sampling (proposal.py:118-143), batched over chains."""
    from .fp import expit_cr, logaddexp_cr

    N, D = state.position.shape
    L = int(num_integration_steps)
    if metric is None:
        metric = default_metric(inverse_mass_matrix, n_chains=N, per_chain_diag=per_chain_diag)
    keys = chain_keys(rng_key, N, chain_offset) if chain_keys_override is None else chain_keys_override
    kk = prng.split(keys, 2)  # hmc.py:299
    key_momentum, key_integrator = kk[:, 0], kk[:, 1]
    p0 = sample_momentum(metric, key_momentum, D)
    z0 = IntegratorState(state.position, p0, state.logdensity, state.logdensity_grad)
    e0 = hmc_energy(metric, z0)  # trajectory.py:211
    prop = [z0.position.copy(), z0.momentum.copy(), z0.logdensity.copy(), z0.logdensity_grad.copy()]
    prop_energy = e0.copy()
    W = np.zeros(N, f32)  # Proposal(initial_state, initial_energy, 0.0, -inf)  trajectory.py:212
    S = np.full(N, -np.inf, f32)
    any_div = np.zeros(N, bool)
    z = z0
    for i in range(L):  # trajectory.py:214-225
        step_keys = prng.fold_in(key_integrator, np.uint32(i))
        z = integrator_step(z, step_size, logdensity_fn, metric, coefficients)
        e_new = hmc_energy(metric, z)
        w = safe_energy_diff(e0, e_new)  # proposal.py:91-95
        s_new = np.minimum(w, f32(0.0))
        any_div |= (-w) > f32(divergence_threshold)
        with np.errstate(invalid="ignore"):
            pa = expit_cr((w - W).astype(f32))  # progressive_uniform_sampling
        acc = prng.uniform(step_keys, ()) < pa
        W = logaddexp_cr(W, w)
        S = logaddexp_cr(S, s_new)
        for a, b in zip(prop, (z.position, z.momentum, z.logdensity, z.logdensity_grad)):
            a[acc] = b[acc]
        prop_energy = np.where(acc, e_new, prop_energy).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        acceptance_rate = (exp_cr(S) / f32(L)).astype(f32)  # hmc.py:234
    proposal = IntegratorState(*prop)
    info = HMCInfo(p0, acceptance_rate, np.ones(N, bool), any_div, prop_energy, proposal, L)
    return HMCState(proposal.position, proposal.logdensity, proposal.logdensity_grad), info


def run(rng_key, state: HMCState, logdensity_fn, step_size, inverse_mass_matrix,
        num_integration_steps, num_steps, divergence_threshold=1000.0, chain_offset=0):
    """util.py:150-213 run_inference_algorithm, step-major keys:
    ``keys = split(rng_key, num_steps)``; step t uses ``split(keys[t], N)``."""
    keys = prng.split(rng_key, num_steps)
    positions, infos = [], []
    for t in range(num_steps):
        state, info = kernel(keys[t], state, logdensity_fn, step_size, inverse_mass_matrix,
                             num_integration_steps, divergence_threshold, chain_offset)
        positions.append(state.position)
        infos.append(info)
    return state, np.stack(positions, axis=0), infos


# ----------------------------------------------------------------------------- dynamic HMC
class DynamicHMCState(NamedTuple):  # blackjax/mcmc/dynamic_hmc.py:39-52
    position: np.ndarray
    logdensity: np.ndarray
    logdensity_grad: np.ndarray
    random_generator_arg: np.ndarray  # (N, 2) uint32: one key per chain


def dynamic_hmc_kernel(rng_key, state: DynamicHMCState, logdensity_fn, step_size,
                       inverse_mass_matrix, divergence_threshold: float = 1000.0,
                       chain_offset: int = 0, steps_bounds=(1, 10), metric=None, multinomial=False,
                       coefficients=None):
    """blackjax/mcmc/dynamic_hmc.py:65-126 with the default callables
    ``integration_steps_fn = lambda key: randint(key, (), 1, 10)`` and
    ``next_random_arg_fn = lambda key: split(key)[1]``: every chain draws its own trajectory
    length from its own ``random_generator_arg``.  Restated chain by chain (small cases)."""
    N, D = state.position.shape
    n_steps = prng.randint(state.random_generator_arg, *steps_bounds)
    keys = chain_keys(rng_key, N, chain_offset)
    eps = np.broadcast_to(np.asarray(step_size, f32), (N,))
    imm = np.asarray(inverse_mass_matrix, f32)
    if metric is None and imm.ndim >= 2 and imm.shape[-1] == imm.shape[-2] and (imm.ndim == 3 or imm.shape[0] != N):
        metric = default_metric(imm, n_chains=N)  # a dense matrix (shared, or one per chain)
    outs = []
    for i in range(N):
        st_i = HMCState(state.position[i:i + 1], state.logdensity[i:i + 1], state.logdensity_grad[i:i + 1])
        imm_i = imm if (imm.ndim == 1 or metric is not None) else imm[i]
        met_i = None
        if metric is not None:  # ``metric``: a prepared (dense) Metric; per-chain matrices are sliced
            met_i = metric
            if metric.is_dense and metric.inverse_mass_matrix.ndim == 3:
                met_i = Metric(metric.inverse_mass_matrix[i], metric.mass_matrix_sqrt[i], True, metric.dense_accum)
        # multinomial=True: blackjax.dmhmc (build_proposal=multinomial_hmc_proposal, __init__.py:155-163)
        outs.append((mhmc_kernel if multinomial else kernel)(
            None, st_i, logdensity_fn, eps[i], imm_i, int(n_steps[i]), divergence_threshold,
            chain_keys_override=keys[i:i + 1], metric=met_i, coefficients=coefficients))
    cat = lambda f: np.concatenate([f(o) for o in outs], 0)
    new = DynamicHMCState(cat(lambda o: o[0].position), cat(lambda o: o[0].logdensity),
                          cat(lambda o: o[0].logdensity_grad),
                          prng.split(state.random_generator_arg, 2)[:, 1])
    info = HMCInfo(cat(lambda o: o[1].momentum), cat(lambda o: o[1].acceptance_rate),
                   cat(lambda o: o[1].is_accepted), cat(lambda o: o[1].is_divergent),
                   cat(lambda o: o[1].energy),
                   IntegratorState(*[cat(lambda o, k=k: o[1].proposal[k]) for k in range(4)]),
                   n_steps)
    return new, info
