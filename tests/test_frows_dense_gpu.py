"""SURVEY.md section 8f rows with DENSE metrics (VERDICT r1 "missing" #3): the reference's
multinomial_hmc_proposal (blackjax/mcmc/hmc.py:181-248), dynamic_hmc (mcmc/dynamic_hmc.py:65-126)
and palindromic integrators (mcmc/integrators.py:335-369) work with any metric; here with one shared
dense matrix (fp32 MFMA GEMMs -> compared bit for bit with the oracle's f32-chain mode) and with one
dense matrix per chain (fp64-accumulated matrix-vector kernels -> the oracle's default mode)."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc
from oracle import integrators as oint
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def _setup(dev, N, D, per_chain, rho=0.8, seed=0):
    """AR(1) target; metric = its covariance (shared) or random SPD matrices (one per chain)."""
    fn_o = otargets.ar1_gaussian(rho, D)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    rng = np.random.default_rng(seed)
    if per_chain:
        a = rng.standard_normal((N, D, D))
        imm = (a @ np.swapaxes(a, 1, 2) / D + 0.5 * np.eye(D)).astype(f32)
        imm = ((imm + np.swapaxes(imm, 1, 2)) * f32(0.5)).astype(f32)
        m_g = bjx.metrics.default_metric(dev_t(imm, dev), N, D, dev)
        metric = ohmc.default_metric(imm, n_chains=N,
                                     mass_matrix_sqrt=np.ascontiguousarray(np.swapaxes(t2n(m_g.mass_sqrt_t), 1, 2)))
        ref = ohmc.default_metric(imm, n_chains=N).mass_matrix_sqrt
    else:
        imm = otargets.ar1_covariance(rho, D)
        m_g = bjx.metrics.default_metric(dev_t(imm, dev), N, D, dev)
        metric = ohmc.default_metric(imm, n_chains=N, dense_accum="f32chain",
                                     mass_matrix_sqrt=np.ascontiguousarray(t2n(m_g.mass_sqrt_t).T))
        ref = ohmc.default_metric(imm, n_chains=N).mass_matrix_sqrt
    np.testing.assert_allclose(metric.mass_matrix_sqrt, ref, rtol=1e-5, atol=1e-6)
    q0 = prng.normal(prng.key(1), (N, D)).astype(f32)
    return fn_o, tgt, imm, metric, q0


@pytest.mark.parametrize("N,D,L,per_chain", [(128, 128, 5, False), (37, 30, 6, False), (20, 24, 5, True), (256, 128, 3, False)])
def test_mhmc_dense_parity(dev, N, D, L, per_chain):
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.mhmc(tgt, 0.3, dev_t(imm, dev), L, chain_offset=2)
    st_g = alg.init(dev_t(q0, dev))
    moved = 0
    for kk in prng.split(prng.key(0), 5):
        st_n, info_o = ohmc.mhmc_kernel(kk, st_o, fn_o, f32(0.3), imm, L, chain_offset=2, metric=metric)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(st_g.position), st_n.position)  # same reservoir picks, same bits
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-6)
        moved += int((st_n.position != st_o.position).any(1).sum())
        st_o = st_n
    assert moved > 0


@pytest.mark.parametrize("N,D,per_chain", [(128, 128, False), (45, 20, False), (18, 16, True), (256, 128, False)])
def test_dynamic_hmc_dense_parity(dev, N, D, per_chain):
    """Per-chain random trajectory lengths through the MASKED dense leapfrog (finished chains are
    copied through untouched by the GEMM's prologue / epilogue)."""
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain, seed=3)
    st = ohmc.init(q0, fn_o)
    rga = prng.split(prng.key(77), N)
    st_o = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, rga)
    alg = bjx.dynamic_hmc(tgt, 0.3, dev_t(imm, dev))
    st_g = alg.init(dev_t(q0, dev), prng.key(77))
    lengths = set()
    for kk in prng.split(prng.key(0), 4):
        st_o, info_o = ohmc.dynamic_hmc_kernel(kk, st_o, fn_o, f32(0.3), imm, metric=metric)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(st_g.position), st_o.position)
        assert np.array_equal(t2n(info_g.proposal.position), info_o.proposal.position)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        lengths |= set(info_o.num_integration_steps.tolist())
    assert len(lengths) >= 4


@pytest.mark.parametrize("N,D,per_chain", [(128, 128, False), (45, 20, False), (18, 16, True), (256, 128, False)])
def test_dmhmc_dense_parity(dev, N, D, per_chain):
    """blackjax.dmhmc with a dense metric (blackjax/__init__.py:155-163; VERDICT r2 "missing" #4): per-chain
    random trajectory lengths AND progressive sampling of one state per trajectory -- the masked dense
    leapfrog + bjx_mhmc_step_dense_masked.  Lengths, reservoir picks (exact positions and momenta) and
    divergence flags follow the oracle."""
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain, seed=7)
    st = ohmc.init(q0, fn_o)
    st_o = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, prng.split(prng.key(77), N))
    alg = bjx.dmhmc(tgt, 0.3, dev_t(imm, dev))
    st_g = alg.init(dev_t(q0, dev), prng.key(77))
    lengths = set()
    moved = 0
    for kk in prng.split(prng.key(0), 4):
        st_n, info_o = ohmc.dynamic_hmc_kernel(kk, st_o, fn_o, f32(0.3), imm, metric=metric, multinomial=True)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert bool(info_g.is_accepted.all())
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert np.array_equal(t2n(st_g.position), st_n.position)  # same reservoir picks, same bits
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-5)
        assert np.array_equal(t2n(st_g.random_generator_arg).view(np.uint32), st_n.random_generator_arg)
        lengths |= set(info_o.num_integration_steps.tolist())
        moved += int((st_n.position != st_o.position).any(1).sum())
        st_o = st_n
    assert len(lengths) >= 4 and moved > 0


@pytest.mark.parametrize("name", ["mclachlan", "yoshida", "omelyan"])
@pytest.mark.parametrize("N,D,per_chain", [(128, 128, False), (33, 18, False), (12, 16, True), (256, 128, False)])
def test_palindromic_integrators_dense_parity(dev, name, N, D, per_chain):
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain, seed=5)
    st_o = ohmc.init(q0, fn_o)
    L = 4
    alg = bjx.hmc(tgt, 0.35, dev_t(imm, dev), L, integrator=getattr(bjx.integrators, name), chain_offset=1)
    st_g = alg.init(dev_t(q0, dev))
    n_rej = 0
    for kk in prng.split(prng.key(2), 4):
        st_o, (p_acc, acc, div, e1, z) = oint.hmc_kernel(kk, st_o, fn_o, f32(0.35), imm, L, getattr(oint, name),
                                                         chain_offset=1, metric=metric)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), acc)
        assert np.array_equal(t2n(info_g.proposal.position), z.position)
        assert np.array_equal(t2n(st_g.position), st_o.position)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), p_acc, rtol=1e-5, atol=1e-7)
        n_rej += int((~acc).sum())
    assert float(info_g.acceptance_rate.mean()) > 0.5


def test_mhmc_dense_window_adaptation_smoke(dev):
    """window_adaptation(mhmc, is_mass_matrix_diagonal=False) then sampling with the adapted per-chain
    dense matrices (reference tests/mcmc/test_sampling.py:317-379 runs this combination)."""
    N, D, L = 64, 4, 10
    tgt = bjx.targets.AR1Gaussian(0.7, D)
    warm = bjx.window_adaptation(bjx.mhmc, tgt, is_mass_matrix_diagonal=False, adaptation_info_fn=None,
                                 num_integration_steps=L)
    (state, params), _ = warm.run(bjx.random.key(1), torch.randn(N, D, device=dev), 150)
    assert params["inverse_mass_matrix"].shape == (N, D, D)
    alg = bjx.mhmc(tgt, **params)
    draws = []
    for k in bjx.random.split(bjx.random.key(2), 60):
        state, info = alg.step(k, state)
        draws.append(state.position)
    x = torch.stack(draws[10:]).reshape(-1, D).double()
    emp = (x.T @ x) / x.shape[0]
    assert float((emp - tgt.covariance(dev).double()).abs().max()) < 0.25


@pytest.mark.parametrize("name", ["mclachlan", "yoshida", "omelyan"])
@pytest.mark.parametrize("eps_pc", [False, True])
@pytest.mark.parametrize("N,D,L,per_chain", [(128, 128, 3, False), (20, 24, 4, True)])
def test_mhmc_dense_general_integrator_parity(dev, name, N, D, L, per_chain, eps_pc):
    """Round 4 (VERDICT r3 "missing" #4): blackjax.mhmc with a DENSE metric and a multi-stage integrator
    (hmc.py:181-248 over integrators.py:335-369) -- every stage a masked-free bjx_leapfrog_dense_coef launch, the
    closing kick b1 + reservoir step bjx_mhmc_step_dense_coef.  Reservoir picks (exact positions and momenta)
    and divergence flags follow the oracle (shared matrix: its f32-chain mode; per chain: fp64 mat-vec)."""
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain, seed=3)
    st_o = ohmc.init(q0, fn_o)
    # eps_pc: one step size PER CHAIN (the closing kick must use eps[chain] * b1: ADVICE r4 found the scalar and the
    # per-chain step size swapped with the coefficient in this call, invisible with a scalar step size)
    eps = np.random.default_rng(11).uniform(0.15, 0.4, N).astype(f32) if eps_pc else f32(0.3)
    alg = bjx.mhmc(tgt, dev_t(eps, dev) if eps_pc else 0.3, dev_t(imm, dev), L, chain_offset=2,
                   integrator=getattr(bjx.integrators, name))
    st_g = alg.init(dev_t(q0, dev))
    moved = 0
    for kk in prng.split(prng.key(0), 3):
        st_n, info_o = ohmc.mhmc_kernel(kk, st_o, fn_o, eps, imm, L, chain_offset=2, metric=metric,
                                        coefficients=getattr(oint, name))
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(st_g.position), st_n.position)
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-5)
        moved += int((st_n.position != st_o.position).any(1).sum())
        st_o = st_n
    assert moved > 0


@pytest.mark.parametrize("name", ["mclachlan", "omelyan"])
@pytest.mark.parametrize("eps_pc", [False, True])
@pytest.mark.parametrize("N,D,per_chain", [(40, 36, False), (14, 12, True)])
def test_dmhmc_dense_general_integrator_parity(dev, name, N, D, per_chain, eps_pc):
    """blackjax.dmhmc (per-chain random trajectory lengths + progressive sampling) with a dense metric and a
    multi-stage integrator: all launches masked by the chain's own length."""
    fn_o, tgt, imm, metric, q0 = _setup(dev, N, D, per_chain, seed=9)
    st = ohmc.init(q0, fn_o)
    st_o = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, prng.split(prng.key(77), N))
    eps = np.random.default_rng(13).uniform(0.15, 0.4, N).astype(f32) if eps_pc else f32(0.3)
    alg = bjx.dmhmc(tgt, dev_t(eps, dev) if eps_pc else 0.3, dev_t(imm, dev),
                    integrator=getattr(bjx.integrators, name))
    st_g = alg.init(dev_t(q0, dev), prng.key(77))
    lengths = set()
    for kk in prng.split(prng.key(0), 3):
        st_n, info_o = ohmc.dynamic_hmc_kernel(kk, st_o, fn_o, eps, imm, metric=metric, multinomial=True,
                                               coefficients=getattr(oint, name))
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert np.array_equal(t2n(st_g.position), st_n.position)
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        lengths |= set(info_o.num_integration_steps.tolist())
        st_o = st_n
    assert len(lengths) >= 3
