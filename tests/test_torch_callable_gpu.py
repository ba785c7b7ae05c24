"""The path north_star names: the user's log-density as a plain PyTorch function (autograd), not a
built-in HIP target.

* parity: the same target as a torch function and as ``bjx.targets.DiagGaussian`` (and the oracle)
  gives identical accept bits and positions within 1e-6 over consecutive transitions -- HMC and NUTS;
* the reference's own non-Gaussian statistical pin restated on the engine
  (/root/reference/tests/mcmc/test_sampling.py:317-379): linear-regression posterior through
  ``window_adaptation(algorithm, logposterior) x {hmc L=90, nuts, mhmc L=20}`` then sampling with
  ``algorithm(logposterior, **parameters)``: mean(scale) ~ 1, mean(coefs) ~ 3, atol 0.1;
* the round trip of the adapted parameters when N == D (a square (N, D) array of
  per-chain diagonals must not be read as a dense matrix).
"""
import math

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def test_autograd_callable_hmc_matches_hip_target_and_oracle(dev):
    N, D, L, T = 300, 192, 12, 6
    sig = (10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(f32)
    imm = (sig * sig).astype(f32)
    inv_var = (f32(1) / imm).astype(f32)
    iv_t = dev_t(inv_var, dev)

    def torch_logdensity(q):  # no gradient returned: goes through torch.autograd
        return -0.5 * (q * q * iv_t).sum(-1)

    q0 = (sig * prng.normal(prng.key(1), (N, D))).astype(f32)
    fn_o = otargets.diag_gaussian(inv_var)
    alg_t = bjx.hmc(torch_logdensity, 0.25, dev_t(imm, dev), L, chain_offset=7)
    alg_h = bjx.hmc(bjx.targets.DiagGaussian(iv_t), 0.25, dev_t(imm, dev), L, chain_offset=7)
    st_t, st_h, st_o = alg_t.init(dev_t(q0, dev)), alg_h.init(dev_t(q0, dev)), ohmc.init(q0, fn_o)
    # autograd's gradient of this function is -(q * inv_var) with one rounding: identical bits
    assert torch.equal(st_t.logdensity_grad, st_h.logdensity_grad)
    n_rej = 0
    for k in prng.split(prng.key(2), T):
        st_t, i_t = alg_t.step(k, st_t)
        st_h, i_h = alg_h.step(k, st_h)
        st_o, i_o = ohmc.kernel(k, st_o, fn_o, f32(0.25), imm, L, chain_offset=7)
        assert torch.equal(i_t.is_accepted, i_h.is_accepted)
        assert np.array_equal(t2n(i_t.is_accepted), i_o.is_accepted)
        np.testing.assert_allclose(t2n(st_t.position), t2n(st_h.position), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(st_t.position), st_o.position, rtol=1e-6, atol=1e-6)
        # torch's fp32 row sum vs the fp64-accumulated one: the only difference between the paths
        np.testing.assert_allclose(t2n(i_t.acceptance_rate), i_o.acceptance_rate, rtol=2e-4, atol=1e-6)
        n_rej += int((~i_o.is_accepted).sum())
    assert n_rej > 0


def test_autograd_callable_nuts_matches_hip_target(dev):
    N, D, T = 128, 24, 5
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(f32)
    iv_t = dev_t((f32(1) / (sig * sig)).astype(f32), dev)

    def torch_logdensity(q):
        return -0.5 * (q * q * iv_t).sum(-1)

    q0 = dev_t((sig * prng.normal(prng.key(3), (N, D))).astype(f32), dev)
    a_t = bjx.nuts(torch_logdensity, 0.2, torch.ones(D, device=dev), max_num_doublings=7)
    a_h = bjx.nuts(bjx.targets.DiagGaussian(iv_t), 0.2, torch.ones(D, device=dev), max_num_doublings=7)
    s_t, s_h = a_t.init(q0), a_h.init(q0)
    for k in prng.split(prng.key(4), T):
        s_t, i_t = a_t.step(k, s_t)
        s_h, i_h = a_h.step(k, s_h)
        assert torch.equal(i_t.num_integration_steps, i_h.num_integration_steps)
        assert torch.equal(i_t.is_turning, i_h.is_turning)
        np.testing.assert_allclose(t2n(s_t.position), t2n(s_h.position), rtol=1e-6, atol=1e-6)
    assert int(i_h.num_integration_steps.max()) > 3


def _regression_logposterior(dev, seed=0):
    """tests/mcmc/test_sampling.py:103-111 (regression_logprob) over a batch of chains:
    position = {"log_scale": (N,), "coefs": (N,)}, 1000 data points, y = 3 x + noise."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(1000, device=dev, generator=g)
    y = 3.0 * x + torch.randn(1000, device=dev, generator=g)
    half_log_2pi = 0.5 * math.log(2.0 * math.pi)

    def logposterior(pos):
        log_scale, coefs = pos["log_scale"], pos["coefs"]
        scale = torch.exp(log_scale)
        scale_prior = -scale + log_scale                       # expon.logpdf(scale) + log_scale
        coefs_prior = -0.5 * (coefs / 5.0) ** 2 - math.log(5.0) - half_log_2pi
        resid = (y[None, :] - coefs[:, None] * x[None, :]) / scale[:, None]
        loglik = (-0.5 * resid * resid - log_scale[:, None] - half_log_2pi).sum(-1)
        return scale_prior + coefs_prior + loglik

    return logposterior


@pytest.mark.parametrize("name,params,n_warm,n_samp", [
    ("hmc", {"num_integration_steps": 90}, 400, 300),
    ("nuts", {}, 400, 200),
    ("mhmc", {"num_integration_steps": 20}, 400, 300),
])
@pytest.mark.parametrize("is_diag", [True, False])
def test_regression_posterior_through_window_adaptation(dev, name, params, n_warm, n_samp, is_diag):
    """The reference's statistical pin (test_sampling.py:317-379) with an autograd log-posterior over a
    dict of parameters, 48 chains at once: every chain warms up on its own, then samples with
    ``algorithm(logposterior, **parameters)``."""
    N = 48
    algorithm = {"hmc": bjx.hmc, "nuts": bjx.nuts, "mhmc": bjx.mhmc}[name]
    tree0 = {"log_scale": torch.zeros(N, device=dev), "coefs": torch.full((N,), 4.0, device=dev)}
    flat0, unravel = bjx.util.ravel_chain_pytree(tree0)
    fn = bjx.util.flat_logdensity(_regression_logposterior(dev), unravel)
    warm = bjx.window_adaptation(algorithm, fn, is_mass_matrix_diagonal=is_diag,
                                 adaptation_info_fn=None, **params)
    (state, parameters), _ = warm.run(bjx.random.key(19), flat0, n_warm)
    alg = algorithm(fn, **parameters)
    draws = []
    for k in bjx.random.split(bjx.random.key(20), n_samp):
        state, info = alg.step(k, state)
        draws.append(state.position)
    tree = unravel(torch.stack(draws).reshape(-1, 2))
    scale_mean = float(torch.exp(tree["log_scale"]).mean())
    coefs_mean = float(tree["coefs"].mean())
    np.testing.assert_allclose(scale_mean, 1.0, atol=1e-1)
    np.testing.assert_allclose(coefs_mean, 3.0, atol=1e-1)
    assert float(info.acceptance_rate.mean()) > 0.5


def test_adapted_parameters_round_trip_when_n_equals_d(dev):
    """N == D: window_adaptation's (N, D) per-chain diagonals come back tagged
    (metrics.PerChainDiagTensor), so ``hmc(fn, **parameters)`` reads them as diagonals, not as one
    dense D x D matrix."""
    N = D = 8
    L = 5
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(f32)
    fn = bjx.targets.DiagGaussian(dev_t((f32(1) / (sig * sig)).astype(f32), dev))
    warm = bjx.window_adaptation(bjx.hmc, fn, num_integration_steps=L, adaptation_info_fn=None)
    (state, parameters), _ = warm.run(bjx.random.key(1), torch.randn(N, D, device=dev), 60)
    imm = parameters["inverse_mass_matrix"]
    assert isinstance(imm, bjx.metrics.PerChainDiagTensor) and imm.shape == (N, D)
    m = bjx.metrics.default_metric(imm, N, D, dev)
    assert m.kind == "diag" and m.imm_stride == D
    try:  # the ambiguity: untagged, the same square array is read as ONE dense matrix (usually not PD)
        assert bjx.metrics.default_metric(imm.as_subclass(torch.Tensor).clone(), N, D, dev).kind == "dense"
    except torch.linalg.LinAlgError:
        pass
    alg = bjx.hmc(fn, **parameters)
    ref = bjx.hmc(fn, parameters["step_size"], bjx.metrics.PerChainDiag(imm.as_subclass(torch.Tensor)), L)
    k = bjx.random.key(2)
    s1, i1 = alg.step(k, state)
    s2, i2 = ref.step(k, state)
    assert torch.equal(s1.position, s2.position) and torch.equal(i1.acceptance_rate, i2.acceptance_rate)
