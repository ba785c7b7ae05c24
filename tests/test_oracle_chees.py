"""Pins the ChEES oracle (oracle/chees.py) on the reference's own tests
(tests/adaptation/test_adaptation.py:77-152, 241-282, 285-310, 313-342, 442-580, 664-755) and checks
the product's host-side scalar logic (blackjax_amd/chees.py, optim.py, dynamic_hmc.halton_sequence)
against it bit for bit.  No GPU needed."""
import numpy as np
import pytest

from oracle import adaptation as oad
from oracle import chees as och
from oracle import hmc as ohmc
from oracle import prng, targets
from oracle.fp import f32, f64


def _gaussian(std):
    std = np.asarray(std, f32)
    return targets.diag_gaussian((f32(1.0) / (std * std)).astype(f32))


# ----------------------------------------------------------------------------- reference pins
def test_chees_statistical_pin_of_the_reference():
    """tests/adaptation/test_adaptation.py:77-152: N(0, diag(1, 100)), 16 chains, 1000 warm-up steps,
    adam(0.5, b1=0, b2=0.95), target 0.75 -> step size 1.5 +- 0.3, mean L 9 +- 3, harmonic-mean
    acceptance 0.75 +- 0.1, sample std within 10 %."""
    std = np.array([1.0, 10.0], f32)
    fn = _gaussian(std)
    k = prng.split(prng.key(346), 3)
    q0 = prng.normal(k[0], (16, 2))
    state, rga, params, _ = och.run(fn, k[1], q0, 0.1, och.Adam(0.5, b1=0, b2=0.95), 1000,
                                    target_acceptance_rate=0.75)
    np.testing.assert_allclose(params["step_size"], 1.5, atol=0.3)
    # sampling phase with the tuned parameters (blackjax.dhmc(**parameters)); one key per step here
    L, jitter = params["integration_steps_params"][0], params["jitter_gn"]
    keys = prng.split(k[2], 500)
    i = int(rga[0])
    inv_acc, steps, draws = [], [], []
    for t in range(500):
        n = och.integration_steps(jitter(i), L)
        state, info = ohmc.kernel(keys[t], state, fn, params["step_size"], params["inverse_mass_matrix"], n)
        i += 1
        steps.append(n)
        inv_acc.append(1.0 / np.mean(1.0 / np.maximum(info.acceptance_rate, 1e-30)))
        draws.append(state.position)
    np.testing.assert_allclose(np.mean(inv_acc), 0.75, atol=0.1)
    np.testing.assert_allclose(np.mean(steps), 9, atol=3)
    draws = np.concatenate(draws, 0)
    np.testing.assert_allclose(draws.mean(0), 0.0, atol=0.5)
    np.testing.assert_allclose(draws.std(0), std, rtol=0.1)


def _random_update_inputs(num_chains, dim, seed=0):
    keys = prng.split(prng.key(seed), 4)
    return (prng.normal(keys[0], (num_chains, dim)), prng.normal(keys[1], (num_chains, dim)),
            prng.normal(keys[2], (num_chains, dim)), prng.uniform(keys[3], (num_chains,)),
            np.zeros(num_chains, bool))


def test_whitened_criterion_reduces_to_raw_when_identity():
    """test_adaptation.py:241-282: bit-for-bit."""
    props, moms, inits, acc, div = _random_update_inputs(8, 4)

    def run_update(whiten):
        init, update = och.base(lambda i: 0.5, lambda i: i + 1, och.Adam(0.5), 0.651, 0.5, 1000, whiten)
        return update(init(0, 0.1), props, moms, inits, acc, div, np.ones(4, f32))

    a, b = run_update(True), run_update(False)
    for x, y in zip(a[:4], b[:4]):
        assert x == y
    assert a.da_state == b.da_state and a.optim_state == b.optim_state


def test_whitened_criterion_correctness():
    """test_adaptation.py:672-733: whitened(props, inits, moms, S) == raw(S^-1/2 props, S^-1/2 inits,
    S^1/2 moms, ones) under SGD(1e-3) (unclipped regime), rtol 1e-6."""
    props, moms, inits, acc, div = _random_update_inputs(12, 4)
    imm = np.array([1e-2, 1e-1, 1e1, 1e2], f32)

    def run_update(whiten, p, m, i, s):
        init, update = och.base(lambda i: 0.5, lambda i: i + 1, och.SGD(1e-3), 0.651, 0.5, 1000, whiten)
        return update(init(0, 0.1), p, m, i, acc, div, s)

    inv_sqrt, sqrt = (1 / np.sqrt(imm)).astype(f32), np.sqrt(imm).astype(f32)
    w = run_update(True, props, moms, inits, imm)
    r = run_update(False, props * inv_sqrt, moms * sqrt, inits * inv_sqrt, np.ones(4, f32))
    np.testing.assert_allclose(w.log_trajectory_length_moving_average,
                               r.log_trajectory_length_moving_average, rtol=1e-6)
    assert abs(w.log_trajectory_length_moving_average - np.log(0.1)) < 0.35  # unclipped


def test_engagement_gate_and_pooled_welford():
    """test_adaptation.py:285-310 + the batch (CGL) merge equals the reference's row-by-row Welford
    fold (chees_adaptation.py:816-824) up to rounding."""
    d = 5
    thr = och.mass_matrix_engagement_threshold(d)
    assert thr >= 64
    below = och.MomentBlock(f32(thr - 1), np.zeros(d, f32), np.zeros(d, f32))
    np.testing.assert_array_equal(och.diagonal_mass_matrix_or_fallback(below, thr, d), np.ones(d, f32))
    scales = np.array([0.1, 1.0, 10.0, 2.0, 5.0], f32)
    samples = (prng.normal(prng.key(0), (thr + 20, d)) * scales).astype(f32)
    block = och.MomentBlock(f32(0.0), np.zeros(d, f32), np.zeros(d, f32))
    for lo in range(0, thr + 20, 21):
        block = och.cgl_update_batch(block, samples[lo:lo + 21])
    mean, m2, n = och.welford_fold_rows(np.zeros(d, f32), np.zeros(d, f32), 0, samples)
    assert n == int(block.count) == thr + 20
    np.testing.assert_allclose(block.mean, mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(block.m2, m2, rtol=1e-5)
    imm = och.diagonal_mass_matrix_or_fallback(block, thr, d)
    assert np.all(np.isfinite(imm)) and not np.allclose(imm, 1.0)
    np.testing.assert_allclose(imm / scales**2, 1.0, rtol=0.5)


def test_mass_matrix_estimation_correctness():
    """test_adaptation.py:313-342: diagonal estimate within rtol 0.6 of the true variances."""
    std = np.array([0.1, 1.0, 10.0], f32)
    k = prng.split(prng.key(2026), 2)
    q0 = (prng.normal(k[0], (32, 3)) * std).astype(f32)
    state, _, params, _ = och.run(_gaussian(std), k[1], q0, 0.1, och.Adam(0.5, b1=0, b2=0.95), 300,
                                  mass_matrix_estimation="diagonal", mass_matrix_window_fraction=0.5)
    np.testing.assert_allclose(params["inverse_mass_matrix"] / std**2, 1.0, rtol=0.6)
    assert np.all(np.isfinite(state.position))


def test_none_matches_omitted_and_floor_inert():
    """test_adaptation.py:162-212."""
    fn = _gaussian([1.0, 10.0])
    q0 = prng.normal(prng.key(7), (16, 2))
    run = lambda **kw: och.run(fn, prng.key(11), q0, 0.1, och.Adam(0.5, b1=0, b2=0.95), 50, **kw)
    (s0, _, p0, _), (s1, _, p1, _) = run(), run(mass_matrix_estimation=None, length_floor=False)
    np.testing.assert_array_equal(s0.position, s1.position)
    assert p0["step_size"] == p1["step_size"]
    assert p0["integration_steps_params"] == p1["integration_steps_params"]


def test_invalid_arguments():
    """test_adaptation.py:215-238."""
    fn = _gaussian([1.0, 1.0])
    q0 = np.zeros((4, 2), f32)
    with pytest.raises(ValueError, match="mass_matrix_estimation"):
        och.run(fn, prng.key(0), q0, 0.1, och.Adam(0.5), 5, mass_matrix_estimation="dense")
    for frac in (1.5, -0.1):
        with pytest.raises(ValueError, match="mass_matrix_window_fraction"):
            och.run(fn, prng.key(0), q0, 0.1, och.Adam(0.5), 5, mass_matrix_estimation="diagonal",
                    mass_matrix_window_fraction=frac)


def test_accumulator_and_power_iteration_recover_planted_eigenvalue():
    """test_adaptation.py:442-480 (NumPy RNG for the correlated draws)."""
    d, rho = 10, 0.9
    C = np.eye(d)
    C[0, 1] = C[1, 0] = rho
    rng = np.random.default_rng(0)
    samples = rng.multivariate_normal(np.zeros(d), C, size=20_000).astype(f32)
    acc = och.MomentBlock(f32(0.0), np.zeros(d, f32), np.zeros((d, d), f32))
    for i in range(100):
        acc = och.cgl_update_batch(acc, samples[i * 200:(i + 1) * 200])
    np.testing.assert_allclose(acc.m2 / (acc.count - 1), np.cov(samples, rowvar=False), atol=2e-5, rtol=2e-5)
    vec = (np.ones(d, f32) / f32(np.sqrt(f32(d)))).astype(f32)
    for _ in range(5):
        vec, lam = och.recompute_eig_state(acc, np.ones(d, f32), vec)
    np.testing.assert_allclose(lam, 1.0 + rho, rtol=0.1)
    assert float(np.sum(vec[2:] ** 2)) < 0.05


def test_power_iteration_converges_from_warm_start():
    """test_adaptation.py:483-504."""
    d, rho = 10, 0.9
    C = np.eye(d, dtype=f32)
    C[0, 1] = C[1, 0] = rho
    v0 = (np.ones(d, f32) / f32(np.sqrt(f32(d)))).astype(f32)
    lam, v = och.power_iteration_lambda_max(C, v0, och.LENGTH_FLOOR_POWER_ITERATIONS)
    np.testing.assert_allclose(lam, 1.0 + rho, rtol=0.05)
    lam2, _ = och.power_iteration_lambda_max(C, v, 1)
    np.testing.assert_allclose(lam2, 1.0 + rho, rtol=1e-3)


def test_apply_length_floor_arithmetic():
    """test_adaptation.py:507-580, 736-755."""
    expected = float(np.pi / 2 * np.sqrt(100.0))
    c, flag = och.apply_length_floor(3.0, 100.0, True, True, 1000, 0.1)
    np.testing.assert_allclose(c, expected, rtol=1e-6)
    assert not flag
    assert och.apply_length_floor(50.0, 100.0, True, True, 1000, 0.1) == (f32(50.0), False)
    assert och.apply_length_floor(3.0, 100.0, True, False) == (f32(3.0), False)
    c, flag = och.apply_length_floor(3.0, 100.0, True, True, 8, 0.1)
    np.testing.assert_allclose(c, 0.8, rtol=1e-6)
    assert flag
    assert och.apply_length_floor(3.0, 1e6, False, True, 1000, 0.1) == (f32(3.0), False)
    np.testing.assert_allclose(och.apply_length_floor(0.0, 4.0, True, True, 1000, 0.1)[0], np.pi, rtol=1e-6)


def test_halton_sequence():
    """dynamic_hmc.py:205-215 + test_adaptation.py:664-669."""
    got = [float(och.halton_sequence(i, 10)) for i in range(8)]
    assert got == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]
    with pytest.raises(ValueError, match="max_bits"):
        och.halton_sequence(0, 32)


# ----------------------------------------------------------------------------- product host logic
def _adam_closed_form(grads, lr, b1, b2, eps, eps_root):
    """optax.scale_by_adam + scale(-lr) written out NON-recursively in fp64 from the documented
    formulas (optax/_src/transform.py: m_t = (1-b1) sum_i b1^(t-i) g_i, v_t likewise with g_i^2,
    bias corrections 1 - b^t, u = m_hat / (sqrt(v_hat + eps_root) + eps)): an independent statement,
    not a copy of the step-by-step restatements in blackjax_amd/optim.py and oracle/chees.py."""
    out = []
    g = np.asarray(grads, f64)
    for t in range(1, len(g) + 1):
        w = np.arange(t - 1, -1, -1, dtype=f64)  # exponents t - i
        m = (1.0 - b1) * np.sum(b1 ** w * g[:t])
        v = (1.0 - b2) * np.sum(b2 ** w * g[:t] ** 2)
        m_hat, v_hat = m / (1.0 - b1 ** t), v / (1.0 - b2 ** t)
        out.append(-lr * m_hat / (np.sqrt(v_hat + eps_root) + eps))
    return np.asarray(out)


def test_optimizers_against_an_independent_closed_form_and_known_answers():
    """Both step-by-step fp32 restatements of optax's Adam / SGD (product and oracle) against (a)
    hand-computed first steps and (b) the closed-form fp64 evaluation over a gradient history."""
    from blackjax_amd import optim

    # known answers: first Adam step is -lr * g / (|g| + eps) whatever b1, b2 (bias correction)
    for mk in (optim.adam, och.Adam):
        opt = mk(0.5, b1=0.9, b2=0.999)
        u, st = opt.update(f32(3.0), opt.init(f32(0.0)), f32(0.0))
        # (1e-5: optax's own fp32 `1 - b2**count` cancels to ~1e-5 relative at count = 1, b2 = 0.999)
        assert abs(float(u) + 0.5) < 1e-5 and st[0] == 1
        u, _ = opt.update(f32(-2.0), opt.init(f32(0.0)), f32(0.0))
        assert abs(float(u) - 0.5) < 1e-5
    for mk in (optim.sgd, och.SGD):
        assert float(mk(1e-3).update(f32(4.0), (), f32(0.0))[0]) == float(f32(-1e-3) * f32(4.0))
    rng = np.random.default_rng(7)
    grads = rng.normal(size=60) * 10.0 ** rng.integers(-3, 3, 60)
    for lr, b1, b2 in ((0.5, 0.0, 0.95), (0.01, 0.9, 0.999), (0.1, 0.5, 0.9)):
        want = _adam_closed_form(grads, lr, b1, b2, 1e-8, 0.0)
        for mk in (optim.adam, och.Adam):
            opt = mk(lr, b1=b1, b2=b2)
            st = opt.init(f32(0.0))
            got = []
            for g in grads:
                u, st = opt.update(f32(g), st, f32(0.0))
                got.append(float(u))
            np.testing.assert_allclose(got, want, rtol=5e-5, atol=1e-9)


def test_product_optimizers_match_the_oracle_bitwise():
    from blackjax_amd import optim

    rng = np.random.default_rng(3)
    grads = np.concatenate([rng.normal(size=50) * 10.0 ** rng.integers(-6, 6, 50), [0.0, np.inf, np.nan]])
    for (mine, ref) in ((optim.adam(0.5, b1=0, b2=0.95), och.Adam(0.5, b1=0, b2=0.95)),
                        (optim.adam(0.01), och.Adam(0.01)), (optim.sgd(1e-3), och.SGD(1e-3))):
        sa, sb = mine.init(f32(0.1)), ref.init(f32(0.1))
        for g in grads:
            ua, sa = mine.update(f32(g), sa, f32(0.0))
            ub, sb = ref.update(f32(g), sb, f32(0.0))
            assert (ua == ub) or (np.isnan(ua) and np.isnan(ub))
            assert tuple(sa) == tuple(sb) or np.isnan(np.asarray(tuple(sa), f64)).any()


def test_product_host_update_matches_the_oracle_bitwise():
    """blackjax_amd.chees.base(...).update.scalar_update fed the four pooled sums reproduces the
    oracle's ChEESAdaptationState exactly, over a sequence of updates (incl. divergent chains and a
    zero acceptance probability)."""
    from blackjax_amd import chees as pch
    from blackjax_amd import optim

    jitter = lambda i: och.halton_sequence(i, 11)
    init_o, update_o = och.base(jitter, lambda i: i + 1, och.Adam(0.5, b1=0, b2=0.95), 0.651, 0.5, 1000)
    init_p, update_p = pch.base(jitter, lambda i: i + 1, optim.adam(0.5, b1=0, b2=0.95), 0.651, 0.5, 1000)
    so, sp = init_o(0, 0.1), init_p(0, 0.1)
    imm = np.ones(5, f32)
    for t in range(30):
        props, moms, inits, acc, div = _random_update_inputs(24, 5, seed=100 + t)
        div[t % 24] = True
        if t == 7:
            acc[3] = 0.0
        nd = ~div
        w = np.where(nd, acc, f32(0.0)).astype(f32)
        crit = och.chain_criterion(props, moms, inits, w, imm, True)
        scale = f32(f32(jitter(so.random_generator_arg)) * so.trajectory_length)
        with np.errstate(divide="ignore"):
            sums = [(f32(1.0) / acc)[nd].astype(f64).sum(), float(nd.sum()),
                    (acc[nd].astype(f64) * (scale * crit).astype(f32)[nd].astype(f64)).sum(),
                    (acc[nd] + f32(1e-20)).astype(f32).astype(f64).sum()]
        so = update_o(so, props, moms, inits, acc, div, imm)
        sp = update_p.scalar_update(sp, sums)
        assert tuple(sp[:4]) == tuple(so[:4]), t
        assert sp.da_state.log_x == so.da_state.log_step_size and sp.da_state.avg_error == so.da_state.avg_error
        assert tuple(sp.optim_state) == tuple(so.optim_state)
        assert (sp.random_generator_arg, sp.step) == (so.random_generator_arg, so.step)


def test_product_halton_and_floor_match_the_oracle():
    from blackjax_amd import chees as pch
    from blackjax_amd.dynamic_hmc import halton_sequence

    for i in range(200):
        assert halton_sequence(i, 11) == och.halton_sequence(i, 11)
    with pytest.raises(ValueError, match="max_bits"):
        halton_sequence(0, 32)
    for args in ((3.0, 100.0, True, True, 1000, 0.1), (3.0, 100.0, True, True, 8, 0.1),
                 (3.0, 1e6, False, True, 1000, 0.1), (3.0, 100.0, True, False, 1000, 0.1)):
        assert pch._apply_length_floor(*args) == och.apply_length_floor(*args)
    assert pch._mass_matrix_engagement_threshold(5) == och.mass_matrix_engagement_threshold(5) == 64
    assert pch._mass_matrix_engagement_threshold(4096) == 128


def test_product_host_uniform_matches_the_oracle():
    from blackjax_amd import random as brandom

    for s in range(20):
        assert brandom.uniform(brandom.key(s)) == prng.uniform(prng.key(s), ())


def test_product_argument_validation():
    from blackjax_amd import chees_adaptation

    with pytest.raises(ValueError, match="mass_matrix_estimation"):
        chees_adaptation(lambda q: q, 4, mass_matrix_estimation="dense")
    for frac in (1.5, -0.1):
        with pytest.raises(ValueError, match="mass_matrix_window_fraction"):
            chees_adaptation(lambda q: q, 4, mass_matrix_estimation="diagonal", mass_matrix_window_fraction=frac)
