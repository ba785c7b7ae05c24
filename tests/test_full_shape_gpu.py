"""HIP path vs oracle AT THE EXACT BASELINE.json SHAPES (configs[1..4]).

Chains are independent and keyed by their GLOBAL chain index, so any subset of the chains of a
full-size run can be recomputed on its own by the oracle: every test here runs the engine on the
whole batch and compares >= 64 chain indices spread over every chain block / workgroup / GEMM row
tile (first, last, block boundaries, random) against the oracle evaluated on just those chains
with ``chain_keys_override`` (``oracle/hmc.py``, ``oracle/nuts.py``, ``oracle/adaptation.py``).
C2 additionally compares ALL 65 536 chains against the oracle's C port (itself bit-identical to the
NumPy oracle, tests/test_oracle_c.py).

Bar: accept bits / tree sizes / divergence and U-turn flags exact; positions exact for the
diagonal metric and -- through the oracle's f32-chain mode -- for the dense metric too.
Reference lines restated by the compared paths: blackjax/mcmc/hmc.py:279-312, nuts.py:113-145,
adaptation/staged_adaptation.py:731-754,860-876.
"""
import types

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import adaptation as oad
from oracle import cport, hmc as ohmc, nuts as onuts
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def spread_indices(N, boundaries, n_random, seed):
    """First/last chains, both sides of every multiple of each boundary size, random fill."""
    idx = {0, 1, 2, 3, 4, 5, 63, 64, N - 2, N - 1}
    for b in boundaries:
        for m in range(b, N, b):
            idx.update((m - 1, m))
    rng = np.random.default_rng(seed)
    idx.update(int(i) for i in rng.choice(N, n_random, replace=False))
    return np.array(sorted(i for i in idx if 0 <= i < N), dtype=np.int64)


def sigma_ladder(D, lo, hi):
    return (10.0 ** (lo + (hi - lo) * np.arange(D) / (D - 1))).astype(f32)


# ------------------------------------------------------------------------------------------- C2
def test_c2_full_shape_all_chains_vs_c_port_and_subset_vs_numpy(dev):
    """configs[1]: HMC diagonal mass, 65 536 chains x 1 024-dim Gaussian, 50 leapfrog steps.
    Three consecutive transitions of the default driver (Infinity-Cache blocks of 16 384 chains):
    every chain against the oracle's C port, 80+ chains also against the NumPy oracle."""
    N, D, L, eps, T = 65536, 1024, 50, 0.25, 3
    sig = sigma_ladder(D, -1.0, 1.0)
    imm = (sig * sig).astype(f32)
    inv_var = (f32(1.0) / imm).astype(f32)
    rng = np.random.default_rng(2)
    q0 = (sig * rng.standard_normal((N, D), dtype=f32)).astype(f32)
    fn_o = otargets.diag_gaussian(inv_var)
    st_c = ohmc.init(q0, fn_o)
    q, lp, g = st_c.position.copy(), st_c.logdensity.copy(), st_c.logdensity_grad.copy()
    idx = spread_indices(N, (16384, 4096), 40, seed=3)
    assert len(idx) >= 64
    st_s = ohmc.HMCState(q[idx].copy(), lp[idx].copy(), g[idx].copy())

    alg = bjx.hmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), eps, dev_t(imm, dev), L)
    # the opt-in engine-resident path (one launch per transition, csrc/bjx_traj.hip) on the same keys: held
    # against the oracle directly, all chains
    alg_f = bjx.hmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), eps, dev_t(imm, dev), L, fuse_target=True)
    st_g = alg.init(dev_t(q0, dev))
    st_f = st_g
    assert np.array_equal(t2n(st_g.logdensity), lp)
    n_rej = 0
    for k in prng.split(prng.key(0), T):
        st_g, info_g = alg.step(k, st_g)
        st_f, info_f = alg_f.step(k, st_f)
        acc, ia, idv = cport.hmc_diag_gaussian_step(k, q, lp, g, eps, imm, inv_var, L)
        assert np.array_equal(t2n(info_f.is_accepted), ia) and np.array_equal(t2n(info_f.acceptance_rate), acc)
        assert np.array_equal(t2n(st_f.position), q) and np.array_equal(t2n(st_f.logdensity), lp)
        assert np.array_equal(t2n(st_f.logdensity_grad), g)
        assert torch.equal(info_f.proposal.momentum, info_g.proposal.momentum)
        st_s, info_s = ohmc.kernel(None, st_s, fn_o, f32(eps), imm, L,
                                   chain_keys_override=prng.split_at(k, idx))
        # every chain vs the C port: decisions, acceptance probabilities, positions, gradients
        assert np.array_equal(t2n(info_g.is_accepted), ia)
        assert np.array_equal(t2n(info_g.is_divergent), idv)
        assert np.array_equal(t2n(info_g.acceptance_rate), acc)
        assert np.array_equal(t2n(st_g.position), q)
        assert np.array_equal(t2n(st_g.logdensity), lp)
        assert np.array_equal(t2n(st_g.logdensity_grad), g)
        # the subset vs the NumPy oracle (and hence NumPy == C port at this shape)
        assert np.array_equal(t2n(info_g.is_accepted)[idx], info_s.is_accepted)
        assert np.array_equal(t2n(st_g.position)[idx], st_s.position)
        assert np.array_equal(t2n(info_g.momentum)[idx], info_s.momentum)
        assert np.array_equal(t2n(info_g.proposal.position)[idx], info_s.proposal.position)
        assert np.array_equal(t2n(info_g.energy)[idx], info_s.energy)
        n_rej += int((~ia).sum())
        assert 0.5 < acc.mean() <= 1.0
    assert 0 < n_rej < N * T


# ------------------------------------------------------------------------------------------- C3
def test_c3_full_shape_nuts_subset_vs_numpy(dev):
    """configs[2]: NUTS (max_depth = 10) on the 256-dim funnel, 32 768 chains.  Five transitions,
    once as a free-running ``run`` and once as lockstep ``step`` calls: tree sizes, depths, flags
    exact and positions equal against the NumPy oracle for 80+ chains spread over the batch PLUS
    the chains that built the deepest trees of the run (picked from the engine's own record --
    only the choice of indices depends on it, the oracle recomputes them from scratch)."""
    N, D, T, eps, depth = 32768, 256, 5, 0.1, 10
    rng = np.random.default_rng(5)
    q0 = (f32(0.1) * rng.standard_normal((N, D), dtype=f32)).astype(f32)
    imm = np.ones(D, f32)
    fn_o = otargets.neal_funnel()

    alg = bjx.nuts(bjx.targets.NealFunnel(), eps, dev_t(imm, dev), max_num_doublings=depth)
    st0 = alg.init(dev_t(q0, dev))
    run_key = prng.key(11)
    keys = prng.split(run_key, T)
    final, positions, rinfo = alg.run(run_key, st0, T)  # free-running chains, step-major keys
    # the opt-in engine-resident path (funnel evaluated inside the multi-tick kernel): the same run, every chain
    final_f, positions_f, rinfo_f = alg.run(run_key, st0, T, fuse_target=True)
    assert torch.equal(positions_f, positions) and torch.equal(final_f.logdensity_grad, final.logdensity_grad)
    for name in ("num_integration_steps", "num_trajectory_expansions", "is_turning", "is_divergent", "energy",
                 "acceptance_rate", "logdensity"):
        assert torch.equal(getattr(rinfo_f, name), getattr(rinfo, name)), name
    total_leaves = rinfo.num_integration_steps.sum(0)
    deepest = np.concatenate([t2n(torch.topk(total_leaves, 8).indices),
                              t2n(torch.topk(rinfo.num_integration_steps.max(0).values, 8).indices)])
    idx = np.unique(np.concatenate([spread_indices(N, (16384, 8192), 50, seed=6), deepest]))
    assert len(idx) >= 64
    st_s = ohmc.init(q0[idx], fn_o)
    st_g = st0
    depths = []
    for t in range(T):
        st_g, info_g = alg.step(keys[t], st_g)
        st_s, info_s = onuts.kernel(None, st_s, fn_o, f32(eps), imm, depth,
                                    chain_keys_override=prng.split_at(keys[t], idx))
        for name in ("num_integration_steps", "num_trajectory_expansions", "is_turning", "is_divergent"):
            want = getattr(info_s, name)
            assert np.array_equal(t2n(getattr(info_g, name))[idx], want), (t, name)
            assert np.array_equal(t2n(getattr(rinfo, name)[t])[idx], want), (t, name, "free-running")
        pos_g = t2n(st_g.position)[idx]
        np.testing.assert_allclose(pos_g, st_s.position, rtol=1e-6, atol=1e-6)
        assert np.mean(pos_g != st_s.position) < 1e-4  # bit-identical up to isolated 1-ulp cases
        np.testing.assert_allclose(t2n(info_g.acceptance_rate)[idx], info_s.acceptance_rate,
                                   rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info_g.energy)[idx], info_s.energy, rtol=1e-6, atol=1e-6)
        # the free-running run is the lockstep run, chain by chain, over ALL chains
        assert torch.equal(positions[t], st_g.position)
        assert torch.equal(rinfo.num_integration_steps[t], info_g.num_integration_steps)
        depths += list(info_s.num_trajectory_expansions)
    assert torch.equal(final.position, st_g.position)
    assert len(set(depths)) >= 3  # trees of several depths among the compared chains
    assert max(depths) == int(rinfo.num_trajectory_expansions.max())  # incl. the deepest tree of the run
    print(f"C3 parity: {len(idx)} chains, tree depths {sorted(set(int(d) for d in depths))}, "
          f"largest tree {int(rinfo.num_integration_steps.max())} leapfrogs")


# ------------------------------------------------------------------------------------------- C4
def test_c4_shard_full_shape_warmup_subset_vs_oracle(dev):
    """configs[3], one GPU's shard: window_adaptation(hmc, L = 50) on the 4 096-dim ill-conditioned
    Gaussian, 32 768 chains with per-chain step size and per-chain inverse mass matrix, as rank 3
    of 8 (chain_offset = 3 x 32 768).  32 warm-up steps: fast buffer, the slow window, ITS END
    (Welford blend + dual-averaging restart) and the closing fast steps.  Per step acceptance
    rates / accept bits / step sizes, and the final positions, step sizes and metrics of 80+
    chains equal the oracle's (NumPy adaptation arithmetic around the C-port transition)."""
    N, D, L, num_steps = 32768, 4096, 50, 32
    off = 3 * N
    sig = sigma_ladder(D, -1.5, 1.5)
    inv_var = (f32(1.0) / (sig * sig)).astype(f32)
    rng = np.random.default_rng(8)
    idx = spread_indices(N, (8192, 3072), 50, seed=9)
    assert len(idx) >= 64
    q0 = torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(10))
    q0_sub = t2n(q0[dev_t(idx, dev)])
    sched = oad.build_schedule(num_steps)
    assert any(e for _, e in sched) and sched[-1] == (0, False)

    def kernel_fn(keys_t, state, step_size, imm):
        q, lp, g = state.position.copy(), state.logdensity.copy(), state.logdensity_grad.copy()
        acc, ia, idv = cport.hmc_diag_gaussian_step_pc(keys_t, q, lp, g, step_size, imm, inv_var, L)
        return ohmc.HMCState(q, lp, g), types.SimpleNamespace(acceptance_rate=acc, is_accepted=ia)

    accepts = []

    def kernel_rec(keys_t, state, step_size, imm):
        st, info = kernel_fn(keys_t, state, step_size, imm)
        accepts.append(info.is_accepted)
        return st, info

    run_key = prng.key(19)
    st_o, par_o, hist_o = oad.window_adaptation_run(
        run_key, q0_sub, otargets.diag_gaussian(inv_var), num_steps, L, kernel_fn=kernel_rec,
        chain_keys_override=prng.split_at(run_key, off + idx))

    keep = bjx.adaptation.get_filter_adapt_info_fn(info_keys={"acceptance_rate", "is_accepted"},
                                                   adapt_state_keys={"step_size"})
    warm = bjx.window_adaptation(bjx.hmc, bjx.targets.DiagGaussian(dev_t(inv_var, dev)),
                                 num_integration_steps=L, adaptation_info_fn=keep)
    (st_g, par_g), info = warm.run(run_key, q0, num_steps, chain_offset=off)
    acc_g = t2n(info.info.acceptance_rate)[:, idx]
    isacc_g = t2n(info.info.is_accepted)[:, idx]
    eps_g = t2n(info.adaptation_state.step_size)[:, idx]
    for t in range(num_steps):
        assert np.array_equal(isacc_g[t], accepts[t]), t
        assert np.array_equal(acc_g[t], hist_o[t][0]), t
        assert np.array_equal(eps_g[t], hist_o[t][1]), t
    assert np.array_equal(t2n(par_g["step_size"])[idx], par_o["step_size"])
    assert np.array_equal(t2n(par_g["inverse_mass_matrix"])[idx], par_o["inverse_mass_matrix"])
    assert np.array_equal(t2n(st_g.position)[idx], st_o.position)
    imm_g = t2n(par_g["inverse_mass_matrix"])[idx]
    assert imm_g.shape == (len(idx), D) and np.ptp(imm_g[:, 0]) > 0  # genuinely per chain
    assert np.ptp(par_o["step_size"]) > 0
    n_acc = int(np.sum(accepts))
    assert 0 < n_acc < num_steps * len(idx)


# ------------------------------------------------------------------------------------------- C5
def test_c5_full_shape_dense_subset_bit_exact(dev):
    """configs[4]: dense mass-matrix HMC on the 512-dim AR(1) Gaussian, 16 384 chains, L = 20, the
    "TN" MFMA GEMM kernel on complete 128 x 128 tiles.  Ten consecutive transitions, no re-sync:
    accept bits, momenta and positions of 170+ chains (at least one in every 128-row GEMM tile)
    are bit-identical to the oracle's f32-chain mode."""
    N, D, L, eps, T, rho = 16384, 512, 20, 0.5, 10, 0.9
    cov = otargets.ar1_covariance(rho, D)
    fn_o = otargets.ar1_gaussian(rho, D)
    rng = np.random.default_rng(12)
    q0 = rng.standard_normal((N, D), dtype=f32)
    idx = np.unique(np.concatenate([spread_indices(N, (4096,), 30, seed=13),
                                    np.arange(0, N, 128) + rng.integers(0, 128, N // 128)]))
    assert len(idx) >= 128 and len(np.unique(idx // 128)) == N // 128
    cov_t = dev_t(cov, dev)
    alg = bjx.hmc(bjx.targets.AR1Gaussian(rho, D), eps, cov_t, L)
    st_g = alg.init(dev_t(q0, dev))
    m = bjx.metrics.default_metric(cov_t, N, D, dev)
    mass_sqrt = np.ascontiguousarray(t2n(m.mass_sqrt_t).T)  # the engine's fp32 factor, L^{-T}
    np.testing.assert_allclose(mass_sqrt, ohmc.default_metric(cov).mass_matrix_sqrt, rtol=2.5e-7, atol=1e-9)
    metric = ohmc.default_metric(cov, dense_accum="f32chain", mass_matrix_sqrt=mass_sqrt)
    st_s = ohmc.init(q0[idx], fn_o)
    assert np.array_equal(t2n(st_g.logdensity_grad)[idx], st_s.logdensity_grad)
    n_rej = 0
    for k in prng.split(prng.key(21), T):
        st_g, info_g = alg.step(k, st_g)
        st_s, info_s = ohmc.kernel(None, st_s, fn_o, f32(eps), cov, L, metric=metric,
                                   chain_keys_override=prng.split_at(k, idx))
        assert np.array_equal(t2n(info_g.momentum)[idx], info_s.momentum)
        assert np.array_equal(t2n(info_g.proposal.position)[idx], info_s.proposal.position)
        assert np.array_equal(t2n(info_g.acceptance_rate)[idx], info_s.acceptance_rate)
        assert np.array_equal(t2n(info_g.is_accepted)[idx], info_s.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent)[idx], info_s.is_divergent)
        assert np.array_equal(t2n(st_g.position)[idx], st_s.position)
        n_rej += int((~info_s.is_accepted).sum())
        assert 0.3 < float(info_g.acceptance_rate.mean()) <= 1.0
    assert n_rej > 0


def test_c5_dense_oracle_own_factor(dev):
    """the test above hands the oracle the ENGINE's Cholesky / triangular-inverse factor, so only the
    GEMM chain is compared bit for bit.  Here the oracle factorises the matrix ITSELF (metrics.py:701-729:
    cholesky + solve_triangular on the host, fp64 rounded once) and runs its fp64-accumulated products; the
    engine uses its own factor and the MFMA fp32 chain.  C5's target and metric (D = 512, AR(1) rho = 0.9, L = 20,
    eps = 0.5), 256 chains (two complete GEMM tile rows), ten transitions without re-sync.  Stated tolerance of
    the two arithmetics per transition: |dq| <= 1e-4, momenta 2e-4, acceptance probability
    5e-4 (exp of an energy difference of a few hundred units carried in fp32); accept bits equal except where the
    uniform draw lies within that 5e-4 of the acceptance probability."""
    N, D, L, eps, T, rho = 256, 512, 20, 0.5, 10, 0.9
    cov = otargets.ar1_covariance(rho, D)
    fn_o = otargets.ar1_gaussian(rho, D)
    q0 = np.random.default_rng(5).standard_normal((N, D), dtype=f32)
    alg = bjx.hmc(bjx.targets.AR1Gaussian(rho, D), eps, dev_t(cov, dev), L)
    st_g = alg.init(dev_t(q0, dev))
    metric_own = ohmc.default_metric(cov)  # the oracle's own factor, fp64-accumulated products
    st_o = ohmc.init(q0, fn_o)
    flips = 0
    for k in prng.split(prng.key(33), T):
        # restart the oracle from the engine's state each transition: a one-ulp difference must not be amplified
        # through ten chaotic trajectories before it is measured
        st_o = ohmc.HMCState(t2n(st_g.position), t2n(st_g.logdensity), t2n(st_g.logdensity_grad))
        st_g, info_g = alg.step(k, st_g)
        st_o, info_o = ohmc.kernel(k, st_o, fn_o, f32(eps), cov, L, metric=metric_own)
        np.testing.assert_allclose(t2n(info_g.momentum), info_o.momentum, rtol=0, atol=2e-5)  # a7: p0 = L^-T z
        np.testing.assert_allclose(t2n(info_g.proposal.position), info_o.proposal.position, rtol=0, atol=1e-4)
        np.testing.assert_allclose(t2n(info_g.proposal.momentum), info_o.proposal.momentum, rtol=0, atol=2e-4)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=0, atol=5e-4)
        diff = t2n(info_g.is_accepted) != info_o.is_accepted
        if diff.any():  # only a uniform draw within the stated tolerance of the acceptance probability may flip
            ki = prng.split(prng.split(k, N), 2)[:, 1]
            u = prng.uniform(ki, ())
            assert np.all(np.abs(u[diff] - info_o.acceptance_rate[diff]) < 5e-4)
            flips += int(diff.sum())
    assert flips <= 2
