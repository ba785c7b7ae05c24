"""effective_sample_size: oracle vs the reference's own test properties
(tests/test_diagnostics.py:42-125 of the reference) and device implementation vs oracle."""
import numpy as np
import pytest
import torch

from blackjax_amd.diagnostics import effective_sample_size as ess_dev
from oracle.diagnostics import effective_sample_size as ess_oracle


@pytest.mark.parametrize("num_chains", [1, 2, 10])
@pytest.mark.parametrize("event_shape", [(), (3,), (5, 7)])
def test_ess_iid_normals(num_chains, event_shape):
    """reference test: iid normals, 5000 draws -> ESS ~ M*T at rtol=10, correct shape."""
    rng = np.random.default_rng(32)
    T = 5000
    x = rng.standard_normal((num_chains, T) + event_shape)
    e = ess_oracle(x)
    assert e.shape == event_shape
    np.testing.assert_allclose(e, num_chains * T, rtol=10)
    assert np.all(e > 0.5 * num_chains * T)
    ed = ess_dev(torch.as_tensor(x)).numpy()
    np.testing.assert_allclose(ed, e, rtol=1e-6)


def test_ess_degenerate_and_antithetic():
    """reference tests 91-125: constant chains -> exactly 0; antithetic +-1 chain -> ESS > T."""
    T = 100
    const = np.ones((4, T))
    assert ess_oracle(const) == 0.0 and float(ess_dev(torch.as_tensor(const))) == 0.0
    per_chain = np.arange(4.0)[:, None] * np.ones((4, T))
    assert ess_oracle(per_chain) == 0.0 and float(ess_dev(torch.as_tensor(per_chain))) == 0.0
    anti = np.tile(np.array([1.0, -1.0]), T // 2)[None]
    assert ess_oracle(anti) > T and float(ess_dev(torch.as_tensor(anti))) > T


@pytest.mark.parametrize("T", [7, 20, 101, 512])
def test_ess_device_matches_oracle_on_correlated_chains(T):
    """AR(1) chains with several correlations incl. negative ones, odd/even T, f32 and f64."""
    rng = np.random.default_rng(T)
    M, E = 6, 9
    phi = np.linspace(-0.8, 0.95, E)
    x = np.zeros((M, T, E))
    x[:, 0] = rng.standard_normal((M, E))
    for t in range(1, T):
        x[:, t] = phi * x[:, t - 1] + np.sqrt(1 - phi**2) * rng.standard_normal((M, E))
    e = ess_oracle(x)
    np.testing.assert_allclose(ess_dev(torch.as_tensor(x)).numpy(), e, rtol=1e-8)
    e32 = ess_dev(torch.as_tensor(x.astype(np.float32))).numpy()
    np.testing.assert_allclose(e32, e, rtol=2e-3)
    # axes arguments
    xt = np.moveaxis(x, (0, 1), (2, 0))  # (T, E, M)
    np.testing.assert_allclose(ess_dev(torch.as_tensor(xt), chain_axis=2, sample_axis=0).numpy(), e, rtol=1e-8)
    if T >= 100:
        assert e[-1] < e[E // 2]  # strongly autocorrelated dimension has the smallest ESS


@pytest.mark.gpu
def test_ess_on_gpu(dev):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8, 300, 16)).cumsum(1) * 0.05 + rng.standard_normal((8, 300, 16))
    e = ess_oracle(x.astype(np.float32))
    ed = ess_dev(torch.as_tensor(x.astype(np.float32), device=dev)).cpu().numpy()
    np.testing.assert_allclose(ed, e, rtol=5e-3)
