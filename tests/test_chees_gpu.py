"""GPU parity of the pooled (cross-chain) ChEES-HMC warmup: the
reductions of include/bjx_pool.h against oracle/chees.py, whole runs of ``chees_adaptation`` against
the oracle, the reference's statistical pin, and size-independent checks at 65 536 chains."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import _lib
from blackjax_amd import chees as pch
from oracle import chees as och
from oracle import hmc as ohmc
from oracle import prng, targets as otargets
from oracle.fp import f32, f64

pytestmark = pytest.mark.gpu


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def _inputs(N, D, seed, poison=True):
    keys = prng.split(prng.key(seed), 6)
    props = (prng.normal(keys[0], (N, D)) * f32(3.0) + f32(1.0)).astype(f32)
    moms = prng.normal(keys[1], (N, D))
    inits = (prng.normal(keys[2], (N, D)) * f32(2.0)).astype(f32)
    acc = prng.uniform(keys[3], (N,))
    div = prng.uniform(keys[4], (N,)) < 0.1
    if poison and N > 8:
        props[2, D // 2] = np.inf  # non-finite proposal row -> weight 0, element masked
        props[5, 0] = np.nan
        inits[3, D - 1] = np.nan  # nanmean skips it
        acc[7] = 0.0
    return props, moms, inits, acc, div


def _oracle_sums(props, moms, inits, acc, div, imm, whiten, scale):
    nd = ~div
    w = np.where(nd, acc, f32(0.0)).astype(f32)
    crit = och.chain_criterion(props, moms, inits, w, imm, whiten)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        tg = (f32(scale) * crit).astype(f32)
        sums = np.array([(f32(1.0) / acc)[nd].astype(f64).sum(), float(nd.sum()),
                         (acc[nd].astype(f64) * tg[nd].astype(f64)).sum(),
                         (acc[nd] + f32(1e-20)).astype(f32).astype(f64).sum()])
    return w, crit, sums


@pytest.mark.parametrize("N,D", [(37, 6), (64, 8), (130, 257), (1000, 64), (3, 1), (77, 20), (41, 100), (9, 128),
                                 (300, 512), (530, 1024), (70, 300), (33, 1028)])
@pytest.mark.parametrize("whiten", [False, True])
@pytest.mark.parametrize("poison", [False, True])
def test_pool_kernels_vs_oracle(dev, N, D, whiten, poison):
    props, moms, inits, acc, div = _inputs(N, D, seed=N + D, poison=poison)
    imm = (10.0 ** np.linspace(-1, 1, D)).astype(f32)
    scale = f32(0.37)
    w_o, crit_o, sums_o = _oracle_sums(props, moms, inits, acc, div, imm, whiten, scale)
    # the finite-row mask is part of weighted_empirical_mean
    finite_rows = np.isfinite(props).all(-1)
    w_o = np.where(finite_rows, w_o, f32(0.0))
    imm_t = dev_t(imm, dev) if whiten else None
    sums = pch._ensemble_scalars(dev_t(props, dev), dev_t(moms, dev), dev_t(inits, dev), dev_t(acc, dev),
                                 dev_t(div, dev), imm_t, scale, None)
    ws = pch._workspace(N, D, dev)
    assert np.array_equal(t2n(ws.w), w_o)
    np.testing.assert_allclose(t2n(ws.pm), och.weighted_empirical_mean(props, np.where(~div, acc, 0).astype(f32)),
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(t2n(ws.im), och.nanmean0(inits), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(t2n(ws.crit), crit_o, rtol=5e-6, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(sums[[0, 1, 3]], sums_o[[0, 1, 3]], rtol=1e-12)
    if np.isfinite(sums_o[2]):
        np.testing.assert_allclose(sums[2], sums_o[2], rtol=1e-5, atol=1e-4)
    else:
        assert not np.isfinite(sums[2])


@pytest.mark.parametrize("N,D", [(37, 6), (1000, 64), (77, 20), (2100, 256), (4200, 512), (9000, 1024), (33, 1028)])
def test_fused_weights_colstats_equals_the_two_launches(dev, N, D):
    """bjx_chees_weights_colstats (one read of q') against bjx_chees_weights + bjx_chees_colstats:
    weights bit for bit and the 4*D fp64 column sums bit for bit (column-owner kernels) or to 1e-13
    (rows-per-wave kernel, 128 < D <= 1024: a different fixed summation order), over every geometry (a
    row inside one wave, whole rows per wave, and the D > 1024 fall-back)."""
    props, moms, inits, acc, div = _inputs(N, D, seed=3 * N + D)
    rows = np.random.default_rng(N).choice(N, size=max(N // 50, 1), replace=False)
    props[rows, (rows * 7) % D] = np.inf
    qp, qi, a_t, d_t = dev_t(props, dev), dev_t(inits, dev), dev_t(acc, dev), dev_t(div, dev)
    ws = pch._workspace(N, D, dev)
    st = _lib.current_stream()
    w1, w2 = torch.full((N,), -1.0, device=dev), torch.full((N,), -1.0, device=dev)
    s1 = torch.empty(4 * D, dtype=torch.float64, device=dev)
    s2 = torch.empty_like(s1)
    _lib.call("bjx_chees_weights", st, N, D, qp.data_ptr(), a_t.data_ptr(), d_t.data_ptr(), w1.data_ptr())
    _lib.call("bjx_chees_colstats", st, N, D, qp.data_ptr(), w1.data_ptr(), qi.data_ptr(), ws.scratch.data_ptr(),
              s1.data_ptr())
    _lib.call("bjx_chees_weights_colstats", st, N, D, qp.data_ptr(), a_t.data_ptr(), d_t.data_ptr(), qi.data_ptr(),
              w2.data_ptr(), ws.scratch.data_ptr(), s2.data_ptr())
    assert torch.equal(w1, w2) and float(w1.min()) >= 0.0
    assert (w1[torch.as_tensor(rows, device=dev)] == 0).all()
    if 128 < D <= 1024 and D % 4 == 0:  # whole rows per wave: another (fixed) summation order of the fp64 sums
        torch.testing.assert_close(s2, s1, rtol=1e-13, atol=1e-11)
    else:
        assert torch.equal(s1, s2)


def test_pool_empty_batch(dev):
    D = 12
    stats = torch.full((4 * D,), 7.0, dtype=torch.float64, device=dev)
    z = torch.zeros(0, D, device=dev)
    _lib.call("bjx_chees_colstats", _lib.current_stream(), 0, D, z.data_ptr(), None, z.data_ptr(), None,
              stats.data_ptr())
    assert float(stats.abs().max()) == 0.0
    out = torch.full((4,), 7.0, dtype=torch.float64, device=dev)
    _lib.call("bjx_chees_scalars", _lib.current_stream(), 0, None, None, None, 1.0, out.data_ptr())
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("N,D", [(50, 7), (256, 64)])
def test_pooled_moment_blocks_vs_oracle(dev, N, D):
    rng = np.random.default_rng(N)
    scales = (10.0 ** rng.uniform(-1, 1, D)).astype(f32)
    blk_o = och.MomentBlock(f32(0.0), np.zeros(D, f32), np.zeros(D, f32))
    cov_o = och.MomentBlock(f32(0.0), np.zeros(D, f32), np.zeros((D, D), f32))
    blk = pch.MomentBlock(f32(0.0), torch.zeros(D, device=dev), torch.zeros(D, device=dev))
    cov = pch.MomentBlock(f32(0.0), torch.zeros(D, device=dev), torch.zeros(D, D, device=dev))
    for t in range(4):
        x = (rng.standard_normal((N, D)) * scales + 5.0).astype(f32)
        blk_o, cov_o = och.cgl_update_batch(blk_o, x), och.cgl_update_batch(cov_o, x)
        blk, cov = pch._cgl_update_diag(blk, dev_t(x, dev), None), pch._cgl_update_dense(cov, dev_t(x, dev), None)
        assert blk.count == blk_o.count
        assert np.array_equal(t2n(blk.mean), blk_o.mean)
        np.testing.assert_allclose(t2n(blk.m2), blk_o.m2, rtol=1e-6)
        np.testing.assert_allclose(t2n(cov.mean), cov_o.mean, rtol=1e-6)
        scale = np.sqrt(np.outer(np.diag(cov_o.m2), np.diag(cov_o.m2)))
        assert np.max(np.abs(t2n(cov.m2) - cov_o.m2) / scale) < 1e-4
    thr = 64
    imm = pch._diagonal_mass_matrix_or_fallback(blk, thr, D)
    np.testing.assert_allclose(t2n(imm), och.diagonal_mass_matrix_or_fallback(blk_o, thr, D), rtol=1e-6)
    assert pch._diagonal_mass_matrix_or_fallback(blk, 10**6, D) is None
    vec_o, lam_o = och.recompute_eig_state(cov_o, t2n(imm), (np.ones(D, f32) / f32(np.sqrt(f32(D)))).astype(f32))
    vec, lam = pch._recompute_eig_state(cov, imm, torch.full((D,), 1.0 / float(np.sqrt(f32(D))), device=dev))
    np.testing.assert_allclose(lam, lam_o, rtol=1e-3)


def test_halton_steps_kernel(dev):
    arg = torch.arange(0, 3000, dtype=torch.int32, device=dev)
    for max_bits, amount, L in [(11, 1.0, 18.647448), (12, 0.7, 3.2), (11, 1.0, 0.4), (20, 0.25, 977.0)]:
        fn = bjx.dynamic_hmc.halton_steps_fn(max_bits, amount)
        got = t2n(fn(arg, L))
        ja, jb = f32(amount), f32(1.0 - amount)
        want = [och.integration_steps(f32(f32(och.halton_sequence(i, max_bits) * ja) + jb), f32(L))
                for i in range(3000)]
        assert np.array_equal(got, np.asarray(want, np.int32))
        assert got[:(1 << max_bits) - 1].min() >= 1  # the radical inverse wraps to 0 at i + 1 = 2^max_bits
    with pytest.raises(ValueError, match="max_bits"):
        bjx.dynamic_hmc.halton_steps_fn(32)


def _gauss(std, dev):
    std = np.asarray(std, f32)
    inv_var = (f32(1.0) / (std * std)).astype(f32)
    return otargets.diag_gaussian(inv_var), bjx.targets.DiagGaussian(dev_t(inv_var, dev))


def _record(t, state, info, adapt):
    return (adapt.step_size, adapt.trajectory_length, info.num_integration_steps,
            info.acceptance_rate.copy(), state.position.copy())


@pytest.mark.parametrize("N,D,kwargs", [
    (48, 6, {}),
    (33, 8, {"jitter_amount": 0.6, "decay_rate": 0.3, "target_acceptance_rate": 0.8}),
    (64, 8, {"mass_matrix_estimation": "diagonal"}),
    (64, 5, {"mass_matrix_estimation": "diagonal", "_length_floor": False, "mass_matrix_window_fraction": 0.2}),
    (40, 4, {"mass_matrix_estimation": "diagonal", "_whiten_criterion": False}),
])
def test_chees_run_matches_oracle(dev, N, D, kwargs):
    """Whole warm-up runs: per-step step size, trajectory length, L, acceptance rates and positions
    follow the oracle (same keys, chain_offset 5)."""
    std = (10.0 ** np.linspace(-0.5, 0.7, D)).astype(f32)
    fn_o, fn_g = _gauss(std, dev)
    q0 = (prng.normal(prng.key(3), (N, D)) * std).astype(f32)
    num_steps = 90
    okw = {k.lstrip("_"): v for k, v in kwargs.items()}
    st_o, rga_o, par_o, (ad_o, hist) = och.run(fn_o, prng.key(11), q0, 0.1, och.Adam(0.5, b1=0, b2=0.95),
                                               num_steps, chain_offset=5, record=_record, **okw)
    warm = bjx.chees_adaptation(fn_g, N, chain_offset=5, **kwargs)
    (st_g, par_g), info = warm.run(prng.key(11), dev_t(q0, dev), 0.1, bjx.optim.adam(0.5, b1=0, b2=0.95),
                                   num_steps)
    eps_g, tl_g = t2n(info.adaptation_state.step_size), t2n(info.adaptation_state.trajectory_length)
    acc_g, L_g = t2n(info.info.acceptance_rate), t2n(info.info.num_integration_steps)
    pos_g = t2n(info.state.position)
    loose = "mass_matrix_estimation" in kwargs  # library GEMM / pooled rounding feed back into the run
    for t in range(num_steps):
        eps_o, tl_o, L_o, acc_o, pos_o = hist[t]
        assert abs(int(L_g[t]) - L_o) <= (1 if loose else 0), t
        np.testing.assert_allclose(eps_g[t], eps_o, rtol=2e-3 if loose else 1e-5, err_msg=str(t))
        np.testing.assert_allclose(tl_g[t], tl_o, rtol=2e-3 if loose else 1e-5, err_msg=str(t))
        if not loose:
            np.testing.assert_allclose(acc_g[t], acc_o, rtol=1e-4, atol=1e-6, err_msg=str(t))
            np.testing.assert_allclose(pos_g[t], pos_o, rtol=1e-4, atol=1e-5, err_msg=str(t))
    np.testing.assert_allclose(par_g["step_size"], par_o["step_size"], rtol=2e-3 if loose else 1e-5)
    np.testing.assert_allclose(par_g["integration_steps_params"][0], par_o["integration_steps_params"][0],
                               rtol=5e-3 if loose else 1e-5)
    np.testing.assert_allclose(t2n(par_g["inverse_mass_matrix"]), par_o["inverse_mass_matrix"],
                               rtol=5e-3 if loose else 0)
    assert np.array_equal(t2n(st_g.random_generator_arg), rga_o)
    assert info.floor_clipped_by_cap == par_o["floor_clipped_by_cap"]


def test_chees_jitter_generator_path(dev):
    N, D = 32, 4
    fn_o, fn_g = _gauss(np.ones(D), dev)
    q0 = prng.normal(prng.key(9), (N, D))
    st_o, _, par_o, (ad_o, hist) = och.run(fn_o, prng.key(21), q0, 0.2, och.Adam(0.5, b1=0, b2=0.95), 40,
                                           jitter_generator=lambda k: prng.uniform(k, ()), record=_record)
    warm = bjx.chees_adaptation(fn_g, N, jitter_generator=bjx.random.uniform,
                                adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn(
                                    set(), {"num_integration_steps"}, {"step_size"}))
    (st_g, par_g), info = warm.run(prng.key(21), dev_t(q0, dev), 0.2, bjx.optim.adam(0.5, b1=0, b2=0.95), 40)
    assert info.state.position is None and info.info.acceptance_rate is None
    assert np.array_equal(t2n(info.info.num_integration_steps), np.asarray([h[2] for h in hist]))
    np.testing.assert_allclose(t2n(info.adaptation_state.step_size), [h[0] for h in hist], rtol=1e-5)
    np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-4, atol=1e-5)
    steps = par_g["integration_steps_fn"](st_g.random_generator_arg, *par_g["integration_steps_params"])
    want = och.integration_steps(par_o["jitter_gn"](40), par_o["integration_steps_params"][0])
    assert set(t2n(steps).tolist()) == {want}


def test_chees_statistical_pin_and_sampling(dev):
    """The reference's own ChEES test (tests/adaptation/test_adaptation.py:77-152) on the engine:
    warm-up, then ``dhmc(**parameters)`` sampling."""
    std = np.array([1.0, 10.0], f32)
    _, fn_g = _gauss(std, dev)
    k = prng.split(prng.key(346), 3)
    q0 = prng.normal(k[0], (16, 2))
    warm = bjx.chees_adaptation(fn_g, 16, target_acceptance_rate=0.75,
                                adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn())
    (last_states, parameters), info = warm.run(k[1], dev_t(q0, dev), 0.1, bjx.optim.adam(0.5, b1=0, b2=0.95), 1000)
    assert info.state.position is None and info.adaptation_state.step_size is None
    np.testing.assert_allclose(parameters["step_size"], 1.5, atol=0.3)
    algorithm = bjx.dhmc(fn_g, **parameters)
    state = last_states
    inv_acc, steps, draws = [], [], []
    for kk in prng.split(k[2], 500):
        state, inf = algorithm.step(kk, state)
        inv_acc.append(1.0 / float((1.0 / inf.acceptance_rate.clamp_min(1e-30)).mean()))
        steps.append(float(inf.num_integration_steps.float().mean()))
        draws.append(t2n(state.position))
    np.testing.assert_allclose(np.mean(inv_acc), 0.75, atol=0.1)
    np.testing.assert_allclose(np.mean(steps), 9, atol=3)
    draws = np.concatenate(draws, 0)
    np.testing.assert_allclose(draws.mean(0), 0.0, atol=0.5)
    np.testing.assert_allclose(draws.std(0), std, rtol=0.1)
    assert int(state.random_generator_arg[0]) == 1500


def test_pool_reductions_full_size(dev):
    """65 536 chains x 256 dims: the pooled statistics against an independent fp64 torch evaluation."""
    N, D = 65536, 256
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    props = torch.randn(N, D, device=dev, generator=g) * 3 + 1
    moms = torch.randn(N, D, device=dev, generator=g)
    inits = torch.randn(N, D, device=dev, generator=g) * 2
    acc = torch.rand(N, device=dev, generator=g)
    div = torch.rand(N, device=dev, generator=g) < 0.05
    imm = torch.logspace(-1, 1, D, device=dev)
    sums = pch._ensemble_scalars(props, moms, inits, acc, div, imm, 0.5, None)
    ws = pch._workspace(N, D, dev)
    w = torch.where(div, torch.zeros_like(acc), acc).double()
    pm = (w[:, None] * props.double()).sum(0) / (w.sum() + 1e-20)
    im = inits.double().mean(0)
    np.testing.assert_allclose(t2n(ws.pm), t2n(pm), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(t2n(ws.im), t2n(im), rtol=2e-6, atol=1e-7)
    isq = 1.0 / imm.double().sqrt()
    pc, ic = (props.double() - pm) * isq, (inits.double() - im) * isq
    vel = moms.double() * imm.double() * isq
    crit = ((pc * pc).sum(1) - (ic * ic).sum(1)) * (pc * vel).sum(1)
    np.testing.assert_allclose(t2n(ws.crit), t2n(crit), rtol=2e-3, atol=2e-1)
    nd = ~div
    want = [float((1.0 / acc[nd].double()).sum()), float(nd.sum()), float((acc[nd].double() * 0.5 * crit[nd]).sum()),
            float((acc[nd].double() + 1e-20).sum())]
    np.testing.assert_allclose(sums[[0, 1, 3]], np.asarray(want)[[0, 1, 3]], rtol=1e-6)
    np.testing.assert_allclose(sums[2], want[2], rtol=1e-3, atol=abs(want[3]) * 1e-2)
    # pooled diagonal block == torch variance
    blk = pch.MomentBlock(f32(0.0), torch.zeros(D, device=dev), torch.zeros(D, device=dev))
    blk = pch._cgl_update_diag(blk, props, None)
    blk = pch._cgl_update_diag(blk, inits, None)
    both = torch.cat([props, inits], 0).double()
    np.testing.assert_allclose(t2n(blk.mean), t2n(both.mean(0)), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(t2n(blk.m2) / (2 * N - 1), t2n(both.var(0)), rtol=1e-5)


@pytest.mark.parametrize("kwargs", [{}, {"mass_matrix_estimation": "diagonal"}])
def test_chees_warmup_with_an_engine_resident_target(dev, kwargs):
    """chees_adaptation(..., fuse_target=True): every warm-up transition is one launch
    (hmc.build_fused_target_kernel); step sizes, trajectory lengths, positions and the returned parameters equal
    the default warm-up's bit for bit."""
    N, D, num_steps = 256, 256, 40
    std = torch.as_tensor((10.0 ** np.linspace(-0.5, 0.7, D)).astype(f32), device=dev)
    fn = bjx.targets.DiagGaussian((1.0 / (std * std)).contiguous())
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    q0 = std * torch.randn(N, D, device=dev, generator=g)
    out = []
    for fuse in (False, True):
        warm = bjx.chees_adaptation(fn, N, fuse_target=fuse, **kwargs)
        (st, par), info = warm.run(prng.key(11), q0, 0.1, bjx.optim.adam(0.5, b1=0, b2=0.95), num_steps)
        out.append((st, par, info))
    (st_a, par_a, info_a), (st_b, par_b, info_b) = out
    assert torch.equal(st_a.position, st_b.position)
    assert par_a["step_size"] == par_b["step_size"]
    assert par_a["integration_steps_params"] == par_b["integration_steps_params"]
    assert torch.equal(par_a["inverse_mass_matrix"], par_b["inverse_mass_matrix"])
    assert torch.equal(info_a.adaptation_state.trajectory_length, info_b.adaptation_state.trajectory_length)
    assert torch.equal(info_a.info.acceptance_rate, info_b.info.acceptance_rate)
