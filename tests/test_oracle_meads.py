"""CPU tests of the GHMC / MEADS oracle (oracle/ghmc.py, oracle/meads.py): the reference's own tests for
this path restated on the oracle -- tests/adaptation/test_meads.py (base(), fold structure, validation)
and tests/mcmc/test_sampling.py::test_ghmc / ::test_meads (statistical pins) -- plus the definition of
``maximum_eigenvalue``.  The ``jax.random`` bit stream stays unpinned (no JAX here, NOTEBOOK.md section 3)."""
import math

import numpy as np

from oracle import ghmc as oghmc
from oracle import meads as omeads
from oracle import prng

f32, f64 = np.float32, np.float64


def _normal_1_2(q):  # stats.norm.logpdf(x, loc=1, scale=2), tests/mcmc/test_sampling.py:1065-1066
    z = ((q[:, 0] - f32(1.0)) / f32(2.0)).astype(f32)
    logp = (f32(-0.5) * z * z - f32(math.log(2.0)) - f32(0.5 * math.log(2.0 * math.pi))).astype(f32)
    return logp, (-(q - f32(1.0)) / f32(4.0)).astype(f32)


def _regression(seed=0, n=1000):
    """tests/mcmc/test_sampling.py:103-111 with position = (coefs, log_scale) (ravel_pytree sorts the
    dict keys), 1 000 points y = 3 x + noise; sufficient statistics make it O(1) per chain."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    y = 3.0 * x + rng.standard_normal(n)
    sxx, sxy, syy = float(x @ x), float(x @ y), float(y @ y)
    c0 = 0.5 * math.log(2.0 * math.pi)

    def fn(q):
        c, ls = q[:, 0].astype(f64), q[:, 1].astype(f64)
        s = np.exp(ls)
        rss = syy - 2.0 * c * sxy + c * c * sxx
        logp = (-s + ls) + (-0.5 * (c / 5.0) ** 2 - math.log(5.0) - c0) + (-0.5 * rss / (s * s) - n * ls - n * c0)
        dc = -c / 25.0 + (sxy - c * sxx) / (s * s)
        dls = -s + 1.0 + rss / (s * s) - n
        return logp.astype(f32), np.stack([dc, dls], axis=1).astype(f32)

    return fn


def test_maximum_eigenvalue_is_the_reference_expression_on_either_gram_matrix():
    rng = np.random.default_rng(0)
    for n, d in ((5, 9), (40, 3), (8, 8)):
        x = rng.standard_normal((n, d)).astype(f32)
        s = x.astype(f64) @ x.astype(f64).T  # meads_adaptation.py:811-817 as written
        diag = np.diag(s)
        want = ((np.sum(s**2) - np.sum(diag**2)) / (n * (n - 1))) / (np.sum(diag) / n)
        np.testing.assert_allclose(omeads.maximum_eigenvalue(x), want, rtol=1e-6)
    # n identical rows v: S = |v|^2 * ones, so the estimate is exactly |v|^2, the one non-zero eigenvalue of X^T X / n
    v = rng.standard_normal(6).astype(f32)
    np.testing.assert_allclose(omeads.maximum_eigenvalue(np.tile(v, (7, 1))), float(v.astype(f64) @ v.astype(f64)), rtol=1e-5)


def test_permutation_is_a_deterministic_permutation():
    p = prng.permutation(prng.key(5), 1000)
    assert sorted(p.tolist()) == list(range(1000)) and not np.array_equal(p, np.arange(1000))
    assert np.array_equal(p, prng.permutation(prng.key(5), 1000))
    assert not np.array_equal(p, prng.permutation(prng.key(6), 1000))
    assert np.array_equal(prng.permutation(prng.key(1), 1), [0])


def test_base_init_and_parameter_effects():
    """tests/adaptation/test_meads.py:29-118 on the oracle."""
    rng = np.random.default_rng(1)
    pos, grads = rng.standard_normal((8, 3)).astype(f32), rng.standard_normal((8, 3)).astype(f32)
    st = omeads.base_init(pos, grads, num_folds=4)
    assert st.step_size.shape == (4,) and st.alpha.shape == (4,) and st.delta.shape == (4,)
    assert st.position_sigma.shape == (4, 3) and st.current_iteration == 0
    for a in (st.step_size, st.alpha, st.delta):
        assert np.all(a == a[0])
    st2 = omeads.base_init(np.ones((8, 3), f32) + 0.1 * pos, np.ones((8, 3), f32), 4, step_size_multiplier=1.0)
    st1 = omeads.base_init(np.ones((8, 3), f32) + 0.1 * pos, np.ones((8, 3), f32), 4, step_size_multiplier=0.5)
    np.testing.assert_allclose(st2.step_size, np.minimum(st1.step_size * 2.0, 1.0), rtol=1e-5)
    hi = omeads.base_init(pos, grads, 4, damping_slowdown=10.0)
    assert np.all(hi.alpha >= st.alpha)
    np.testing.assert_allclose(st.delta, st.alpha / 2)


def test_ghmc_samples_the_univariate_normal():
    """tests/mcmc/test_sampling.py:1160-1172 (test_ghmc): step 1.0, scale 1.0, alpha 0.8, delta 2.0 on
    N(1, 2^2); 6 000 transitions, 1 000 burn-in, mean and variance within 10 % -- here for 8
    independent chains at once."""
    N = 8
    state = oghmc.init(np.ones((N, 1), f32), _normal_1_2, prng.key(3))
    draws = []
    for k in prng.split(prng.key(19), 6000):
        state, info = oghmc.kernel(k, state, _normal_1_2, 1.0, 1.0, 0.8, 2.0)
        draws.append(state.position[:, 0].copy())
    x = np.asarray(draws[1000:])
    np.testing.assert_allclose(x.mean(), 1.0, rtol=1e-1)
    np.testing.assert_allclose(x.var(), 4.0, rtol=1e-1)
    assert np.all(np.abs(state.slice) <= 1.0) and info.num_integration_steps == 1


def test_ghmc_rejection_flips_the_refreshed_momentum_and_keeps_the_state():
    N, D = 64, 5
    fn = lambda q: ((-0.5 * (q.astype(f64) ** 2).sum(1)).astype(f32), (-q).astype(f32))  # noqa: E731
    state = oghmc.init(prng.normal(prng.key(1), (N, D)), fn, prng.key(2))
    new, info = oghmc.kernel(prng.key(4), state, fn, 1.9, np.full(D, 1.0, f32), 0.3, 0.15)
    rej = ~info.is_accepted
    assert rej.any() and info.is_accepted.any()
    assert np.array_equal(new.position[rej], state.position[rej])
    assert np.array_equal(new.momentum[rej], -info.momentum[rej])
    assert np.array_equal(new.slice[rej], (((state.slice + f32(1)) + f32(0.15)) % f32(2) - f32(1))[rej])
    acc = info.is_accepted
    assert np.array_equal(new.position[acc], info.proposal.position[acc])
    assert np.array_equal(new.momentum[acc], -info.proposal.momentum[acc])


def test_meads_fold_structure_and_regression_posterior():
    """tests/mcmc/test_sampling.py:606-690 (test_meads): 128 chains, 4 folds, 1 000 warm-up steps on the
    regression posterior; frozen folds; per-fold step sizes finite and positive; 100 GHMC transitions
    per chain with the adapted parameters recover scale = 1 and coef = 3 (atol 0.1)."""
    fn = _regression()
    N, K, n = 128, 4, 32
    init = np.stack([4.0 + prng.normal(prng.key(11), (N,)), 1.0 + prng.normal(prng.key(12), (N,))], axis=1).astype(f32)
    states, params, hist = omeads.run(prng.key(19), init, fn, 1000, num_folds=K)
    p0 = hist[0][0].position
    assert np.array_equal(p0[:n], init[:n])  # step 0: fold 0 frozen
    assert np.array_equal(hist[1][0].position[n:2 * n], p0[n:2 * n])
    assert np.array_equal(hist[2][0].position[2 * n:3 * n], hist[1][0].position[2 * n:3 * n])
    assert not np.array_equal(p0[n:], init[n:])
    eps = np.stack([h[1].step_size for h in hist])
    assert eps.shape == (1000, K) and np.all(np.isfinite(eps)) and np.all(eps > 0)
    assert params["momentum_inverse_scale"].shape == (2,) and np.ndim(params["step_size"]) == 0
    draws = []
    state = states
    for k in prng.split(prng.key(20), 100):
        state, _ = oghmc.kernel(k, state, fn, **params)
        draws.append(state.position.copy())
    x = np.asarray(draws)
    np.testing.assert_allclose(x[..., 0].mean(), 3.0, atol=1e-1)
    np.testing.assert_allclose(np.exp(x[..., 1]).mean(), 1.0, atol=1e-1)


def test_meads_single_fold_never_freezes():
    """tests/adaptation/test_meads.py:192-224."""
    fn = lambda q: ((-0.5 * (q.astype(f64) ** 2).sum(1)).astype(f32), (-q).astype(f32))  # noqa: E731
    q0 = prng.normal(prng.key(1), (8, 2))
    states, params, hist = omeads.run(prng.key(2), q0, fn, 5, num_folds=1)
    assert not np.allclose(states.position, q0)
    for t in range(1, 5):
        assert not np.allclose(hist[t][0].position, hist[t - 1][0].position)
    assert np.ndim(params["step_size"]) == 0


def test_dynamic_multinomial_hmc_oracle_structure():
    """blackjax.dmhmc on the oracle (tests/mcmc/test_multinomial_hmc.py:151-199): per-chain trajectory
    lengths from randint(key, 1, 10), ``is_accepted`` always True, the acceptance-rate diagnostic is
    exp(sum_log_p_accept) / L of the chain's own L, and the next random_generator_arg is split(key)[1]."""
    from oracle import hmc as ohmc

    N, D = 12, 4
    fn = lambda q: ((-0.5 * (q.astype(f64) ** 2).sum(1)).astype(f32), (-q).astype(f32))  # noqa: E731
    st = ohmc.init(prng.normal(prng.key(1), (N, D)), fn)
    rga = prng.split(prng.key(7), N)
    st = ohmc.DynamicHMCState(st.position, st.logdensity, st.logdensity_grad, rga)
    new, info = ohmc.dynamic_hmc_kernel(prng.key(3), st, fn, f32(0.2), np.ones(D, f32), multinomial=True)
    assert info.is_accepted.all() and not info.is_divergent.any()
    assert np.array_equal(info.num_integration_steps, prng.randint(rga, 1, 10))
    assert len(set(info.num_integration_steps.tolist())) > 2
    assert np.all(info.acceptance_rate > 0.5) and np.all(info.acceptance_rate <= 1.0 + 1e-6)  # fp32 rounding of exp(S) / L
    assert np.array_equal(new.random_generator_arg, prng.split(rga, 2)[:, 1])
    # the endpoint proposal on the same keys takes the same trajectory lengths but (generally) another state
    new_e, info_e = ohmc.dynamic_hmc_kernel(prng.key(3), st, fn, f32(0.2), np.ones(D, f32))
    assert np.array_equal(info_e.num_integration_steps, info.num_integration_steps)
    assert not np.array_equal(new_e.position, new.position)


def test_product_permutation_matches_the_oracle():
    """blackjax_amd.meads._permutation (host keys from the library's bjx_keys_split) against
    oracle/prng.py::permutation -- two restatements of jax.random.permutation."""
    from blackjax_amd.meads import _permutation

    for seed, n in ((5, 1), (5, 2), (6, 128), (7, 1000), (8, 65536)):
        assert np.array_equal(_permutation(prng.key(seed), n), prng.permutation(prng.key(seed), n)), (seed, n)
