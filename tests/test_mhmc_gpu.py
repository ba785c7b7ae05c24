"""GPU parity of multinomial HMC (blackjax.mhmc) vs the oracle."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


@pytest.mark.parametrize("N,D,L,per_chain", [(64, 256, 12, False), (9, 37, 7, True), (5, 8, 1, False)])
def test_mhmc_parity(dev, N, D, L, per_chain):
    rng = np.random.default_rng(N + D)
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / max(D - 1, 1))).astype(np.float32)
    inv_var = (np.float32(1) / (sig * sig)).astype(np.float32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(np.float32) if per_chain else (sig * sig).astype(np.float32)
    eps = rng.uniform(0.05, 0.3, N).astype(np.float32) if per_chain else np.float32(0.2)
    fn_o = otargets.diag_gaussian(inv_var)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.mhmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)),
                   dev_t(eps, dev) if per_chain else float(eps), dev_t(imm, dev), L, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev))
    kept_initial = 0
    for kk in prng.split(prng.key(0), 5):
        st_o_new, info_o = ohmc.mhmc_kernel(kk, st_o, fn_o, eps, imm, L, chain_offset=3)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(st_g.position), st_o_new.position)  # same reservoir picks
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        assert np.array_equal(t2n(st_g.logdensity_grad), st_o_new.logdensity_grad)
        np.testing.assert_allclose(t2n(st_g.logdensity), st_o_new.logdensity, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6, atol=1e-6)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert bool(info_g.is_accepted.all())
        kept_initial += int(np.all(st_o_new.position == st_o.position, axis=1).sum())
        st_o = st_o_new
    if L == 1:
        assert kept_initial > 0  # the "never replaced the initial proposal" path was exercised


def test_mhmc_statistics_and_window_adaptation(dev):
    """reference tests/mcmc/test_sampling.py:317-379 runs window_adaptation x mhmc (L = 20)."""
    N, D, L = 512, 8, 20
    sig = np.array([0.1, 0.3, 1, 3, 0.5, 2, 1, 0.2], np.float32)
    fn = bjx.targets.DiagGaussian(dev_t(1 / (sig * sig), dev))
    warm = bjx.window_adaptation(bjx.mhmc, fn, adaptation_info_fn=None, num_integration_steps=L)
    (state, params), _ = warm.run(bjx.random.key(3), torch.randn(N, D, device=dev), 200)
    alg = bjx.mhmc(fn, params["step_size"], bjx.metrics.PerChainDiag(params["inverse_mass_matrix"]), L)
    draws = []
    for k in bjx.random.split(bjx.random.key(5), 30):
        state, info = alg.step(k, state)
        draws.append(state.position)
    x = torch.stack(draws[5:]).reshape(-1, D)
    np.testing.assert_allclose(t2n(x.var(0)), sig * sig, rtol=0.2)
    assert abs(float(x.mean())) < 0.2
