"""GPU parity of dynamic HMC (per-chain random trajectory length)."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import _lib

dynamic_hmc = bjx.dynamic_hmc
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def test_per_chain_key_kernels(dev):
    N = 1000
    keys = prng.split(prng.key(5), N)
    kt = dev_t(keys.view(np.int32), dev)
    nxt = dynamic_hmc.next_key_fn(kt)
    assert np.array_equal(t2n(nxt).view(np.uint32), prng.split(keys, 2)[:, 1])
    for lo, hi in [(1, 10), (0, 1), (5, 5), (3, 1000), (-7, 9)]:
        r = dynamic_hmc.randint_steps_fn(kt, lo, hi)
        assert np.array_equal(t2n(r), prng.randint(keys, lo, hi))


def test_dynamic_hmc_parity(dev):
    N, D = 40, 24
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(np.float32)
    inv_var = (np.float32(1) / (sig * sig)).astype(np.float32)
    imm = (sig * sig).astype(np.float32)
    fn_o = otargets.diag_gaussian(inv_var)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32)
    st = ohmc.init(q0, fn_o)
    rga = prng.split(prng.key(77), N)
    st_o = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, rga)
    alg = bjx.dynamic_hmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), 0.3, dev_t(imm, dev))
    st_g = alg.init(dev_t(q0, dev), prng.key(77))
    assert np.array_equal(t2n(st_g.random_generator_arg).view(np.uint32), rga)
    lengths = set()
    for kk in prng.split(prng.key(0), 4):
        st_o, info_o = ohmc.dynamic_hmc_kernel(kk, st_o, fn_o, np.float32(0.3), imm)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(st_g.position), st_o.position)
        assert np.array_equal(t2n(st_g.random_generator_arg).view(np.uint32), st_o.random_generator_arg)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.proposal.position), info_o.proposal.position, rtol=1e-6, atol=1e-6)
        lengths |= set(info_o.num_integration_steps.tolist())
    assert len(lengths) >= 5 and min(lengths) >= 1 and max(lengths) <= 9


def test_dynamic_hmc_statistics(dev):
    N, D = 2048, 8
    sig = np.array([0.1, 0.3, 1, 3, 0.5, 2, 1, 0.2], np.float32)
    fn = bjx.targets.DiagGaussian(dev_t(1 / (sig * sig), dev))
    alg = bjx.dynamic_hmc(fn, 0.5, dev_t(sig * sig, dev), integration_steps_params=(1, 16))
    state = alg.init(dev_t(sig, dev) * torch.randn(N, D, device=dev), bjx.random.key(3))
    for k in bjx.random.split(bjx.random.key(5), 40):
        state, info = alg.step(k, state)
    x = state.position
    np.testing.assert_allclose(t2n(x.var(0)), sig * sig, rtol=0.2)
    assert int(info.num_integration_steps.max()) <= 15 and int(info.num_integration_steps.min()) >= 1


@pytest.mark.parametrize("N,D,per_chain", [(40, 24, False), (23, 130, True), (9, 5, True)])
def test_dmhmc_parity(dev, N, D, per_chain):
    """blackjax.dmhmc (blackjax/__init__.py:155-163; tests/mcmc/test_multinomial_hmc.py:151-199): every
    chain draws its own trajectory length and one state of its trajectory by progressive sampling.
    Trajectory lengths, divergence flags and the sampled states follow the oracle; is_accepted is
    always True; the state type is DynamicHMCState."""
    rng = np.random.default_rng(N + D)
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(np.float32)
    inv_var = (np.float32(1) / (sig * sig)).astype(np.float32)
    imm = (sig * sig * (rng.uniform(0.5, 2.0, (N, D)) if per_chain else 1.0)).astype(np.float32)
    eps = rng.uniform(0.1, 0.5, N).astype(np.float32) if per_chain else np.float32(0.3)
    fn_o = otargets.diag_gaussian(inv_var)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32)
    st = ohmc.init(q0, fn_o)
    st_o = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, prng.split(prng.key(77), N))
    alg = bjx.dmhmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), dev_t(eps, dev) if per_chain else 0.3,
                    bjx.metrics.PerChainDiag(dev_t(imm, dev)) if per_chain else dev_t(imm, dev))
    st_g = alg.init(dev_t(q0, dev), prng.key(77))
    assert type(st_g).__name__ == "DynamicHMCState"
    lengths = set()
    for kk in prng.split(prng.key(0), 4):
        st_o, info_o = ohmc.dynamic_hmc_kernel(kk, st_o, fn_o, eps, imm, multinomial=True)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert bool(info_g.is_accepted.all())
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-5)
        assert np.array_equal(t2n(st_g.random_generator_arg).view(np.uint32), st_o.random_generator_arg)
        lengths |= set(info_o.num_integration_steps.tolist())
    assert len(lengths) >= 4
    # the explicit build_proposal spelling is the same kernel (test_multinomial_hmc.py:177-197)
    from blackjax_amd.hmc import multinomial_hmc_proposal

    kern = bjx.dynamic_hmc.build_kernel(build_proposal=multinomial_hmc_proposal)
    a, _ = kern(prng.key(9), st_g, bjx.targets.DiagGaussian(dev_t(inv_var, dev)),
                dev_t(eps, dev) if per_chain else 0.3,
                bjx.metrics.PerChainDiag(dev_t(imm, dev)) if per_chain else dev_t(imm, dev))
    b, _ = alg.step(prng.key(9), st_g)
    assert torch.equal(a.position, b.position)


def test_dmhmc_statistics_and_validation(dev):
    sig = np.array([0.1, 0.3, 1, 3, 0.5, 2, 1, 0.2], np.float32)
    fn = bjx.targets.DiagGaussian(dev_t(1 / (sig * sig), dev))
    alg = bjx.dmhmc(fn, 0.5, dev_t(sig * sig, dev), integration_steps_params=(1, 16))
    state = alg.init(dev_t(sig, dev) * torch.randn(2048, 8, device=dev), bjx.random.key(3))
    for k in bjx.random.split(bjx.random.key(5), 40):
        state, info = alg.step(k, state)
    np.testing.assert_allclose(t2n(state.position.var(0)), sig * sig, rtol=0.2)
    # dense metric x multi-stage integrator: built in round 4 (oracle parity: tests/test_frows_dense_gpu.py)
    st2, info2 = bjx.dmhmc(fn, 0.5, torch.eye(8, device=dev), integrator=bjx.integrators.mclachlan).step(
        bjx.random.key(1), state)
    assert bool(torch.isfinite(st2.position).all()) and bool(info2.is_accepted.all())
    with pytest.raises(NotImplementedError):
        bjx.dynamic_hmc.build_kernel(build_proposal=lambda *a: None)
