"""Pins the oracle's deterministic arithmetic against the reference's own golden vectors
(tests/golden/reference_kats.json, transcribed from /root/reference/tests with file:line)."""
import json
import os

import numpy as np

from oracle import hmc as ohmc
from oracle.fp import f32, f64, fma32

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_fma32_matches_libm():
    import ctypes

    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    rng = np.random.default_rng(0)
    a = rng.standard_normal(20000).astype(f32)
    b = rng.standard_normal(20000).astype(f32)
    c = (-(a.astype(f64) * b)).astype(f32) * f32(1 + 1e-3)
    r = fma32(a, b, c)
    ref = np.array([libm.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], dtype=f32)
    assert np.array_equal(r, ref)


def test_velocity_verlet_mvnormal_golden():
    k = KATS["velocity_verlet_mvnormal"]
    cov = np.array(k["cov"], dtype=f32)
    P = np.linalg.inv(cov.astype(f64))

    def fn(q):
        g = -(q.astype(f64) @ P.T)
        return (0.5 * np.sum(q * g, -1)).astype(f32), g.astype(f32)

    q = np.array([k["q_init"]], dtype=f32)
    p = np.array([k["p_init"]], dtype=f32)
    metric = ohmc.default_metric(cov)
    assert metric.is_dense
    lp, g = fn(q)
    z = ohmc.IntegratorState(q, p, lp, g)
    e0 = ohmc.hmc_energy(metric, z)
    for _ in range(k["num_steps"]):
        z = ohmc.velocity_verlet(z, k["step_size"], fn, metric)
    np.testing.assert_allclose(z.position[0], k["q_final"], atol=k["atol"])
    np.testing.assert_allclose(z.momentum[0], k["p_final"], atol=k["atol"])
    assert abs(float(ohmc.hmc_energy(metric, z)[0] - e0[0])) < 1e-4


def test_velocity_verlet_analytic_examples():
    """tests/mcmc/test_integrators.py:105-135: free fall (g=1), harmonic oscillator, Kepler."""
    k = KATS["velocity_verlet_analytic"]

    def free_fall(q):  # FreeFall(g=1): logdensity = -q, gradient = -1
        return (-q[:, 0]).astype(f32), -np.ones_like(q)

    def harmonic(q):
        return (-0.5 * q[:, 0] ** 2).astype(f32), (-q).astype(f32)

    def kepler(q):
        r2 = np.sum(q.astype(f64) ** 2, -1)
        return (1.0 / np.sqrt(r2)).astype(f32), (-q / (r2[:, None] ** 1.5)).astype(f32)

    for name, fn in [("free_fall", free_fall), ("harmonic_oscillator", harmonic),
                     ("planetary_motion", kepler)]:
        e = k[name]
        q = np.array([e["q_init"]], dtype=f32)
        p = np.array([e["p_init"]], dtype=f32)
        metric = ohmc.default_metric(np.array(e["imm"], dtype=f32))
        lp, g = fn(q)
        z = ohmc.IntegratorState(q, p, lp, g)
        e0 = ohmc.hmc_energy(metric, z)
        for _ in range(e["num_steps"]):
            z = ohmc.velocity_verlet(z, e["step_size"], fn, metric)
        np.testing.assert_allclose(z.position[0], e["q_final"], atol=k["position_atol"])
        if name != "free_fall":  # the reference only asserts the position (its p_final is not checked)
            np.testing.assert_allclose(z.momentum[0], e["p_final"], atol=k["position_atol"])
        assert abs(float(ohmc.hmc_energy(metric, z)[0] - e0[0])) < k["energy_atol"]


def test_hmc_kernel_statistics_normal():
    """tests/mcmc/test_sampling.py:1055-1187 flavour: N(1, 2^2) target, hmc eps=3.9/ L=30 is the
    reference's univariate setting; here a small multi-chain run checks mean/var to rtol 0.1."""
    from oracle import prng

    def fn(q):
        g = -(q - f32(1.0)) / f32(4.0)
        return (0.5 * np.sum((q - f32(1.0)).astype(f64) * g, -1)).astype(f32), g.astype(f32)

    N, D = 256, 1
    st = ohmc.init(np.ones((N, D), f32), fn)
    keys = prng.split(prng.key(12), 60)
    draws = []
    for t, kk in enumerate(keys):
        st, info = ohmc.kernel(kk, st, fn, f32(1.0), np.ones(D, f32) * 4, 5)
        if t >= 20:
            draws.append(st.position.copy())
    d = np.concatenate(draws)
    np.testing.assert_allclose(d.mean(), 1.0, atol=0.1)
    np.testing.assert_allclose(d.var(), 4.0, rtol=0.1)


def test_mhmc_oracle_statistics_normal():
    """Multinomial HMC restatement (hmc.py:181-248): N(1, 2^2) target, mean/var within 0.1/10 %."""
    from oracle import prng

    def fn(q):
        g = -(q - f32(1.0)) / f32(4.0)
        return (0.5 * np.sum((q - f32(1.0)).astype(f64) * g, -1)).astype(f32), g.astype(f32)

    N = 256
    st = ohmc.init(np.ones((N, 1), f32), fn)
    draws = []
    for t, kk in enumerate(prng.split(prng.key(12), 60)):
        st, info = ohmc.mhmc_kernel(kk, st, fn, f32(0.8), np.ones(1, f32) * 4, 6)
        assert info.is_accepted.all()
        if t >= 20:
            draws.append(st.position.copy())
    d = np.concatenate(draws)
    np.testing.assert_allclose(d.mean(), 1.0, atol=0.1)
    np.testing.assert_allclose(d.var(), 4.0, rtol=0.1)
