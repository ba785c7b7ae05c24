"""Rank-normalised split-chain diagnostics (blackjax/diagnostics.py:92-155, 404-522): the torch
implementation against the NumPy/SciPy oracle, and the properties the reference's own tests assert
(tests/test_diagnostics.py:128-440: shapes, iid calibration, non-convergence detection, axis
invariance).  arviz is not installed here, so the arviz calibration tests are not restated."""
import numpy as np
import pytest
import torch

from blackjax_amd import diagnostics as D
from oracle import diagnostics as OD
from oracle import prng

NCHAINS, NSAMPLES = 4, 2000


def _iid(seed, shape=(NCHAINS, NSAMPLES)):
    return prng.normal(prng.key(seed), shape).astype(np.float64)


@pytest.mark.parametrize("fn", ["rhat", "ess_bulk", "ess_tail"])
def test_matches_oracle(fn):
    rng = np.random.default_rng(5)
    # AR(1) chains with different means/scales per event dimension, odd number of draws
    T, M, E = 601, 6, 5
    x = np.zeros((M, T, E))
    eps = rng.standard_normal((M, T, E))
    for t in range(1, T):
        x[:, t] = 0.7 * x[:, t - 1] + eps[:, t]
    x = x * np.array([0.1, 1, 10, 1, 1]) + np.array([0, 0, 0, 5, -5])
    x[3:, :, 3] += 2.0  # un-mixed dimension
    got = getattr(D, fn)(torch.as_tensor(x)).numpy()
    want = getattr(OD, fn)(x)
    np.testing.assert_allclose(got, want, rtol=1e-6)
    got32 = getattr(D, fn)(torch.as_tensor(x.astype(np.float32))).numpy()
    np.testing.assert_allclose(got32, want, rtol=5e-3)
    # axes arguments
    xt = np.moveaxis(x, (0, 1), (2, 0))  # (T, E, M)
    got_t = getattr(D, fn)(torch.as_tensor(xt), chain_axis=2, sample_axis=0).numpy()
    np.testing.assert_allclose(got_t, want, rtol=1e-6)
    got_neg = getattr(D, fn)(torch.as_tensor(xt), chain_axis=-1, sample_axis=-3).numpy()
    np.testing.assert_allclose(got_neg, want, rtol=1e-6)


def test_rhat_properties_of_the_reference_tests():
    x = torch.as_tensor(_iid(13))
    assert D.rhat(x).shape == ()
    assert D.rhat(torch.as_tensor(_iid(13, (NCHAINS, NSAMPLES, 5)))).shape == (5,)
    assert abs(float(D.rhat(x)) - 1.0) < 0.05
    means = np.array([0.0, 5.0, -5.0, 10.0])[:, None]
    assert float(D.rhat(torch.as_tensor(_iid(14) + means))) > 1.1
    # scale non-convergence: same mean, very different variances -- the folded component catches it
    scale = np.array([0.1, 0.1, 10.0, 10.0])[:, None]
    mixed = torch.as_tensor(_iid(15) * scale)
    assert float(D.rhat(mixed)) > 1.05
    assert float(D.potential_scale_reduction(mixed)) < 1.05  # the classic R-hat misses it
    np.testing.assert_allclose(float(D.rhat(x)), float(D.rhat(x.T, chain_axis=1, sample_axis=0)), rtol=1e-5)
    np.testing.assert_allclose(float(D.rhat(x)), float(D.rhat(x, chain_axis=-2, sample_axis=-1)), rtol=1e-5)


def test_ess_bulk_and_tail_properties_of_the_reference_tests():
    total = NCHAINS * NSAMPLES
    x = torch.as_tensor(_iid(7))
    eb = float(D.ess_bulk(x))
    assert D.ess_bulk(x).shape == () and 0.5 * total < eb < 2.0 * total
    assert D.ess_bulk(torch.as_tensor(_iid(7, (NCHAINS, NSAMPLES, 3)))).shape == (3,)
    t = np.arange(NSAMPLES, dtype=np.float64)
    stuck = np.broadcast_to(np.sin(2 * np.pi * t / NSAMPLES)[None], (NCHAINS, NSAMPLES)).copy()
    assert float(D.ess_bulk(torch.as_tensor(stuck))) < eb
    np.testing.assert_allclose(eb, float(D.ess_bulk(x.T, chain_axis=1, sample_axis=0)), rtol=1e-5)
    y = torch.as_tensor(_iid(99))
    et = float(D.ess_tail(y))
    assert D.ess_tail(y).shape == () and 0.2 * total < et < 2.0 * total
    assert D.ess_tail(torch.as_tensor(_iid(99, (NCHAINS, NSAMPLES, 3)))).shape == (3,)
    np.testing.assert_allclose(et, float(D.ess_tail(y.T, chain_axis=1, sample_axis=0)), rtol=1e-5)
    # a wider central interval looks further into the tails: fewer exceedances, different ESS
    assert float(D.ess_tail(y, prob=0.5)) != et


@pytest.mark.gpu
def test_rank_diagnostics_on_gpu(dev):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((64, 400, 16)).cumsum(1) * 0.05 + rng.standard_normal((64, 400, 16))
    xt = torch.as_tensor(x.astype(np.float32), device=dev)
    for fn in ("rhat", "ess_bulk", "ess_tail"):
        got = getattr(D, fn)(xt).cpu().numpy()
        np.testing.assert_allclose(got, getattr(OD, fn)(x), rtol=1e-2)
