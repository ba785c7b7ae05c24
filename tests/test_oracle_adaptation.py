"""Pins the oracle's warmup arithmetic against the reference's own tests
(tests/golden/reference_kats.json: schedules, dual-averaging fixed point, Welford recovery)."""
import itertools
import json
import os

import numpy as np
import pytest

from oracle import adaptation as oad
from oracle import hmc as ohmc
from oracle import prng, targets
from oracle.fp import f32

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.mark.parametrize("num_steps", ["19", "100", "200"])
def test_build_schedule_golden(num_steps):
    expected = []
    for stage, end, count in KATS["build_schedule"][num_steps]:
        expected += [(stage, end)] * count
    got = oad.build_schedule(int(num_steps))
    assert len(got) == int(num_steps)
    assert [(int(s), bool(e)) for s, e in got] == expected


@pytest.mark.parametrize("num_steps", ["19", "100", "200"])
def test_product_build_schedule_golden(num_steps):
    """The PRODUCT's own host schedule (blackjax_amd.adaptation.build_schedule) on the reference's
    vectors (tests/adaptation/test_adaptation.py:27-49) -- not only through the warm-up parity tests --
    and equal to the oracle's on a sweep of lengths."""
    from blackjax_amd.adaptation import build_schedule

    expected = []
    for stage, end, count in KATS["build_schedule"][num_steps]:
        expected += [(stage, end)] * count
    got = build_schedule(int(num_steps))
    assert [(int(s), bool(e)) for s, e in got] == expected
    for n in (0, 1, 20, 21, 149, 150, 151, 333, 1000, 2500):
        assert [(int(s), bool(e)) for s, e in build_schedule(n)] == [(int(s), bool(e)) for s, e in oad.build_schedule(n)]


def test_dual_averaging_golden():
    k = KATS["dual_averaging"]
    st = oad.da_init(np.array([k["x_init"]], f32))
    for _ in range(k["num_updates"]):
        x = np.exp(st.log_step_size)
        g = 2 * (x - 1)  # grad of (x-1)^2
        st = oad.da_update(st, g, gamma=k["gamma"])
    assert abs(float(oad.da_final(st)[0]) - k["expected_final"]) < k["delta"]


@pytest.mark.parametrize("n_dim,is_diag", list(itertools.product([1, 3], [True, False])))
def test_welford_golden(n_dim, is_diag):
    k = KATS["welford"]
    np.random.seed(k["numpy_seed"])
    mu = np.random.randn(n_dim)
    a = np.random.randn(n_dim, n_dim)
    cov = np.matmul(a.T, a)
    samples = np.random.multivariate_normal(mu, cov, k["num_samples"])
    mm = oad.MassMatrixState(np.ones((1, n_dim), f32) if is_diag else np.eye(n_dim, dtype=f32)[None],
                             oad.welford_init(1, n_dim, is_diag))
    for s in samples:
        mm = oad.MassMatrixState(mm.inverse_mass_matrix,
                                 oad.welford_update(mm.wc_state, s[None].astype(f32), is_diag))
    est = oad.mm_final(mm, is_diag).inverse_mass_matrix[0]
    np.testing.assert_allclose(est, np.diagonal(cov) if is_diag else cov, rtol=k["rtol"])
    assert oad.mm_final(mm, is_diag).wc_state.sample_size == 0  # Welford state is reset


def test_window_adaptation_recovers_scales():
    """tests/mcmc/test_sampling.py:317-379 flavour, per-chain adaptation on a diagonal Gaussian:
    the adapted inverse mass matrix tracks sigma^2 and the acceptance rate settles near 0.8."""
    N, D, L = 6, 8, 8
    sig = np.array([0.1, 0.3, 1, 3, 0.5, 2, 1, 0.2], f32)
    fn = targets.diag_gaussian((1 / sig**2).astype(f32))
    q0 = prng.normal(prng.key(2), (N, D)) * sig
    st, params, hist = oad.window_adaptation_run(prng.key(19), q0.astype(f32), fn, 400, L)
    assert params["step_size"].shape == (N,) and params["inverse_mass_matrix"].shape == (N, D)
    ratio = params["inverse_mass_matrix"] / sig**2
    assert np.all(ratio > 0.3) and np.all(ratio < 3.0)
    acc_tail = np.mean([h[0] for h in hist[-50:]])
    assert 0.6 < acc_tail < 0.95
    assert np.all(params["step_size"] > 0.05) and np.all(params["step_size"] < 3)
