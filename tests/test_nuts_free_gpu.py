"""Free-running (asynchronous) NUTS chains (include/bjx_nuts.h, blackjax_amd.nuts.run_free): many
transitions per chain without lockstep must reproduce, chain by chain and transition by transition,
what the oracle's (and the engine's) lockstep transitions produce with the same keys."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc
from oracle import nuts as onuts
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
ATOL = 1e-6
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def _check_against_oracle(dev, fn_o, fn_g, q0, eps, imm, T, max_depth, key_layout, chain_offset=0):
    N, D = q0.shape
    run_key = prng.key(42)
    alg = bjx.nuts(fn_g, dev_t(eps, dev) if np.ndim(eps) else float(eps),
                   bjx.metrics.PerChainDiag(dev_t(imm, dev)) if np.ndim(imm) == 2 else dev_t(imm, dev),
                   max_num_doublings=max_depth, chain_offset=chain_offset)
    st_g0 = alg.init(dev_t(q0, dev))
    final, positions, info = alg.run(run_key, st_g0, T, key_layout=key_layout)
    assert positions.shape == (T, N, D)
    st_o = ohmc.init(q0, fn_o)
    depths = []
    for t in range(T):
        if key_layout == "step_major":
            st_o, info_o = onuts.kernel(prng.split(run_key, T)[t], st_o, fn_o, eps, imm, max_depth,
                                        chain_offset=chain_offset, per_chain_diag=np.ndim(imm) == 2)
        else:
            ck = prng.split(prng.split(run_key, N, offset=chain_offset), T)[:, t]
            st_o, info_o = onuts.kernel(None, st_o, fn_o, eps, imm, max_depth, chain_keys_override=ck,
                                        per_chain_diag=np.ndim(imm) == 2)
        assert np.array_equal(t2n(info.num_integration_steps[t]), info_o.num_integration_steps), t
        assert np.array_equal(t2n(info.num_trajectory_expansions[t]), info_o.num_trajectory_expansions)
        assert np.array_equal(t2n(info.is_divergent[t]), info_o.is_divergent)
        assert np.array_equal(t2n(info.is_turning[t]), info_o.is_turning)
        np.testing.assert_allclose(t2n(positions[t]), st_o.position, rtol=ATOL, atol=ATOL)
        np.testing.assert_allclose(t2n(info.logdensity[t]), st_o.logdensity, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(info.acceptance_rate[t]), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(t2n(info.energy[t]), info_o.energy, rtol=1e-6, atol=1e-6)
        depths += list(info_o.num_trajectory_expansions)
    np.testing.assert_allclose(t2n(final.position), st_o.position, rtol=ATOL, atol=ATOL)
    np.testing.assert_allclose(t2n(final.logdensity_grad), st_o.logdensity_grad, rtol=ATOL, atol=ATOL)
    return depths


@pytest.mark.parametrize("key_layout", ["step_major", "chain_major"])
@pytest.mark.parametrize("N,D", [(24, 16), (7, 5)])
def test_free_running_matches_oracle_gaussian(dev, key_layout, N, D):
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(f32)
    depths = _check_against_oracle(dev, otargets.diag_gaussian(inv_var),
                                   bjx.targets.DiagGaussian(dev_t(inv_var, dev)), q0, f32(0.15),
                                   np.ones(D, f32), T=5, max_depth=7, key_layout=key_layout, chain_offset=3)
    assert len(set(depths)) > 2  # chains were at different depths: the schedule really was asynchronous


def test_free_running_matches_oracle_funnel_per_chain_params(dev):
    """Neal's funnel with per-chain step sizes and per-chain diagonal metrics, divergences and
    max-depth stops included."""
    N, D = 20, 8
    rng = np.random.default_rng(3)
    q0 = (0.5 * prng.normal(prng.key(2), (N, D))).astype(f32)
    eps = rng.uniform(0.05, 0.6, N).astype(f32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(f32)
    depths = _check_against_oracle(dev, otargets.neal_funnel(), bjx.targets.NealFunnel(), q0, eps, imm,
                                   T=4, max_depth=5, key_layout="step_major")
    assert max(depths) == 5


def test_free_running_equals_lockstep_steps_at_scale(dev):
    """4 096 chains x 64 dims, funnel: run(T) == T x step, bit for bit (positions, tree sizes)."""
    N, D, T = 4096, 64, 6
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=8)
    st0 = alg.init(q0)
    final, positions, info = alg.run(bjx.random.key(9), st0, T)
    st = st0
    for t, k in enumerate(bjx.random.split(bjx.random.key(9), T)):
        st, inf = alg.step(k, st)
        assert torch.equal(info.num_integration_steps[t], inf.num_integration_steps), t
        assert torch.equal(info.is_divergent[t], inf.is_divergent)
        assert torch.equal(positions[t], st.position), t
        assert torch.equal(info.acceptance_rate[t], inf.acceptance_rate)
    assert torch.equal(final.position, st.position) and torch.equal(final.logdensity, st.logdensity)
    assert int(info.num_trajectory_expansions.max()) >= 6
    # store_positions=False returns only the final state and the scalar records
    final2, none_pos, info2 = alg.run(bjx.random.key(9), st0, T, store_positions=False)
    assert none_pos is None and torch.equal(final2.position, final.position)
    assert torch.equal(info2.energy, info.energy)
    # HIP-graph replay of the tick chunks (and the batch compactions in between) changes nothing
    alg_g = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=8,
                     use_graph=True)
    final3, pos3, info3 = alg_g.run(bjx.random.key(9), st0, T)
    assert torch.equal(pos3, positions) and torch.equal(final3.logdensity_grad, final.logdensity_grad)
    assert torch.equal(info3.num_integration_steps, info.num_integration_steps)


def test_run_inference_algorithm_free_running_matches_step_loop(dev):
    """blackjax/util.py:150-213 over nuts.step == the same call with free_running=True."""
    N, D, T = 40, 12, 9
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    fn = bjx.targets.NealFunnel()
    alg = bjx.nuts(fn, 0.2, torch.ones(D, device=dev), max_num_doublings=5)
    q0 = 0.3 * torch.randn(N, D, device=dev, generator=g)
    for layout in ("step_major", "chain_major"):
        st_a, (states, infos) = bjx.util.run_inference_algorithm(prng.key(3), alg, T, initial_position=q0,
                                                                 key_layout=layout)
        st_b, (positions, rinfo) = bjx.util.run_inference_algorithm(prng.key(3), alg, T, initial_position=q0,
                                                                    key_layout=layout, free_running=True)
        assert torch.equal(st_a.position, st_b.position)
        assert torch.equal(states.position, positions)
        assert torch.equal(infos.acceptance_rate, rinfo.acceptance_rate)
        assert torch.equal(infos.is_divergent, rinfo.is_divergent)


def test_free_running_is_shard_invariant(dev):
    """Chains sharded over ranks (contiguous blocks, chain_offset = first global chain index) give the
    draws of the unsharded run: free-running NUTS and the free-running warm-up need no exchange."""
    N, D, T = 48, 10, 6
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    q0 = 0.4 * torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.NealFunnel()
    imm = torch.ones(D, device=dev)
    for layout in ("step_major", "chain_major"):
        full = bjx.nuts(fn, 0.25, imm, max_num_doublings=5)
        st, pos, info = full.run(prng.key(21), full.init(q0), T, key_layout=layout)
        parts = []
        for lo, hi in ((0, 20), (20, 48)):
            shard = bjx.nuts(fn, 0.25, imm, max_num_doublings=5, chain_offset=lo)
            _, p, i = shard.run(prng.key(21), shard.init(q0[lo:hi].contiguous()), T, key_layout=layout)
            parts.append((p, i))
        assert torch.equal(torch.cat([p for p, _ in parts], dim=1), pos)
        assert torch.equal(torch.cat([i.num_integration_steps for _, i in parts], dim=1), info.num_integration_steps)
    warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, initial_step_size=0.3, max_num_doublings=5)
    (s_full, p_full), _ = warm.run(prng.key(5), q0, 40, free_running=True)
    outs = [warm.run(prng.key(5), q0[lo:hi].contiguous(), 40, chain_offset=lo, free_running=True)[0]
            for lo, hi in ((0, 20), (20, 48))]
    assert torch.equal(torch.cat([o.state.position for o in outs]), s_full.position)
    assert torch.equal(torch.cat([o.parameters["step_size"] for o in outs]), p_full["step_size"])


@pytest.mark.parametrize("N,D", [(40, 384), (33, 512), (20, 260)])
def test_free_running_two_rows_per_lane(dev, N, D):
    """256 < D <= 512: the low-traffic tick kernels with two 16-byte pieces per lane (NI = 2);
    D = 260 has a ragged second piece.  run(T) == T x step and == the oracle on a few chains."""
    T = 4
    rng = np.random.default_rng(D)
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(f32)
    eps = rng.uniform(0.05, 0.2, N).astype(f32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(f32)
    alg = bjx.nuts(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), dev_t(eps, dev),
                   bjx.metrics.PerChainDiag(dev_t(imm, dev)), max_num_doublings=6)
    st0 = alg.init(dev_t(q0, dev))
    final, positions, info = alg.run(prng.key(8), st0, T)
    st = st0
    for t, k in enumerate(prng.split(prng.key(8), T)):
        st, inf = alg.step(k, st)
        assert torch.equal(info.num_integration_steps[t], inf.num_integration_steps), t
        assert torch.equal(positions[t], st.position), t
        assert torch.equal(info.acceptance_rate[t], inf.acceptance_rate)
    fn_o = otargets.diag_gaussian(inv_var)
    idx = np.array([0, 1, N // 2, N - 1])
    st_o = ohmc.init(q0[idx], fn_o)
    for t, k in enumerate(prng.split(prng.key(8), T)):
        st_o, info_o = onuts.kernel(None, st_o, fn_o, eps[idx], imm[idx], 6, per_chain_diag=True,
                                    chain_keys_override=prng.split_at(k, idx))
        assert np.array_equal(t2n(info.num_integration_steps[t])[idx], info_o.num_integration_steps)
        np.testing.assert_allclose(t2n(positions[t])[idx], st_o.position, rtol=ATOL, atol=ATOL)


@pytest.mark.parametrize("per_chain,N,D", [(False, 14, 9), (True, 14, 9), (False, 40, 70), (True, 11, 130)])
def test_free_running_dense_metric_equals_lockstep_and_oracle(dev, per_chain, N, D):
    """Free-running chains with a dense inverse mass matrix (shared (D, D) or one per chain): the tick
    kernel draws p = L^{-T} z and v = M^{-1} p per chain at the start of ITS transition and runs the
    dense leaf / merge arithmetic of the lockstep kernels, so run(T) == T x step bit for bit; tree
    sizes and positions follow the oracle (metrics.py:260-304, nuts.py:113-145)."""
    T, rho, depth = 4, 0.7, 6
    fn_o = otargets.ar1_gaussian(rho, D)
    rng = np.random.default_rng(D)
    if per_chain:
        a = rng.standard_normal((N, D, D))
        imm = (a @ np.swapaxes(a, 1, 2) / D + 0.5 * np.eye(D)).astype(f32)
    else:
        imm = otargets.ar1_covariance(rho, D)
    q0 = prng.normal(prng.key(6), (N, D)).astype(f32)
    alg = bjx.nuts(bjx.targets.AR1Gaussian(rho, D), 0.3, dev_t(imm, dev), max_num_doublings=depth)
    st0 = alg.init(dev_t(q0, dev))
    run_key = prng.key(8)
    final, positions, info = alg.run(run_key, st0, T)
    st_g, st_o = st0, ohmc.init(q0, fn_o)
    for t, k in enumerate(prng.split(run_key, T)):
        st_g, inf = alg.step(k, st_g)
        assert torch.equal(info.num_integration_steps[t], inf.num_integration_steps), t
        assert torch.equal(positions[t], st_g.position), t
        assert torch.equal(info.acceptance_rate[t], inf.acceptance_rate)
        assert torch.equal(info.energy[t], inf.energy)
        assert torch.equal(info.is_turning[t], inf.is_turning)
        if D <= 16:  # the oracle's dense NUTS is a Python loop per leaf
            st_o, info_o = onuts.kernel(k, st_o, fn_o, f32(0.3), imm, depth)
            assert np.array_equal(t2n(info.num_integration_steps[t]), info_o.num_integration_steps)
            np.testing.assert_allclose(t2n(positions[t]), st_o.position, rtol=1e-5, atol=1e-5)
    assert torch.equal(final.position, st_g.position) and torch.equal(final.logdensity_grad, st_g.logdensity_grad)
    assert len(set(t2n(info.num_trajectory_expansions).ravel().tolist())) > 1
    # recorded tick chunks change nothing
    alg_g = bjx.nuts(bjx.targets.AR1Gaussian(rho, D), 0.3, dev_t(imm, dev), max_num_doublings=depth, use_graph=True)
    final_g, pos_g, info_g = alg_g.run(run_key, st0, T)
    assert torch.equal(pos_g, positions) and torch.equal(info_g.num_integration_steps, info.num_integration_steps)


@pytest.mark.parametrize("target,N,D,T", [("funnel", 9000, 256, 4), ("funnel", 300, 320, 6), ("gauss", 700, 256, 5),
                                          ("funnel", 40, 64, 8)])
def test_fused_target_ticks_equal_the_external_callable_path(dev, target, N, D, T):
    """``run(..., fuse_target=True)``: the tick kernels evaluate the library's own target themselves (one
    launch per tick, bjx_nuts_async_t.target_kind) with the device function the stand-alone target kernel
    runs -- every record, position and the final state are bit for bit those of the default path (two
    launches per tick).  9 000 rows: the multi-tick kernel over a large batch; the small cases: the same kernel on the
    recorded tail's fixed-capacity buffers."""
    import blackjax_amd as bjx

    g = torch.Generator(device=dev)
    g.manual_seed(3)
    if target == "funnel":
        fn = bjx.targets.NealFunnel()
        q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
        eps = 0.15
    else:
        iv = (0.5 + torch.rand(D, device=dev, generator=g)).contiguous()
        fn = bjx.targets.DiagGaussian(iv)
        q0 = torch.randn(N, D, device=dev, generator=g)
        eps = 0.3
    alg = bjx.nuts(fn, eps, torch.ones(D, device=dev), max_num_doublings=6)
    st0 = alg.init(q0)
    key = bjx.random.key(11)
    st_a, pos_a, info_a = alg.run(key, st0, T)
    st_b, pos_b, info_b = alg.run(key, st0, T, fuse_target=True)
    assert torch.equal(pos_a, pos_b)
    for a, b in zip(st_a, st_b):
        assert torch.equal(a, b)
    for name in ("logdensity", "acceptance_rate", "energy", "num_integration_steps", "num_trajectory_expansions",
                 "is_divergent", "is_turning"):
        assert torch.equal(getattr(info_a, name), getattr(info_b, name)), name
    if target == "funnel":
        assert int(info_a.num_integration_steps.max()) > int(info_a.num_integration_steps.min())


def test_fused_target_is_refused_where_it_does_not_apply(dev):
    import blackjax_amd as bjx

    fn = lambda q: -0.5 * (q * q).sum(-1)  # noqa: E731  (not a library target)
    alg = bjx.nuts(fn, 0.2, torch.ones(8, device=dev), max_num_doublings=3)
    st = alg.init(torch.zeros(4, 8, device=dev))
    with pytest.raises(NotImplementedError):
        alg.run(bjx.random.key(0), st, 2, fuse_target=True)
    small = bjx.nuts(bjx.targets.DiagGaussian(torch.ones(64, device=dev)), 0.2, torch.ones(64, device=dev))
    with pytest.raises(NotImplementedError):  # D <= 128: the stand-alone kernel reduces in another order
        small.run(bjx.random.key(0), small.init(torch.zeros(4, 64, device=dev)), 2, fuse_target=True)


@pytest.mark.parametrize("N,D,depth", [(600, 256, 7), (9000, 128, 5)])
def test_step_through_the_free_running_engine_equals_the_lockstep_step(dev, N, D, depth):
    """``nuts(..., fuse_target=True).step``: one free-running transition per chain (key_layout="step") instead
    of the lockstep tree.  State and every scalar info field are bit for bit those of the default ``step``,
    for a plain key and for a ChainMajorKey; the trajectory ends and the momentum are not reported."""
    import blackjax_amd as bjx

    g = torch.Generator(device=dev)
    g.manual_seed(5)
    fn = bjx.targets.NealFunnel()
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    imm = torch.ones(D, device=dev)
    ref = bjx.nuts(fn, 0.15, imm, max_num_doublings=depth)
    fused = bjx.nuts(fn, 0.15, imm, max_num_doublings=depth, fuse_target=True)
    sa = sb = ref.init(q0)
    for t, key in enumerate([bjx.random.key(3), bjx.random.key(4), bjx.random.ChainMajorKey(bjx.random.key(9), 2)]):
        sa, ia = ref.step(key, sa)
        sb, ib = fused.step(key, sb)
        for a, b in zip(sa, sb):
            assert torch.equal(a, b), t
        for name in ("is_divergent", "is_turning", "energy", "num_trajectory_expansions", "num_integration_steps",
                     "acceptance_rate"):
            assert torch.equal(getattr(ia, name), getattr(ib, name)), (t, name)
        assert ib.momentum is None and ib.trajectory_leftmost_state is None
    assert int(ia.num_integration_steps.max()) > int(ia.num_integration_steps.min())


def test_fused_target_random_shapes(dev):
    """Engine-resident NUTS targets on row lengths that are not multiples of 256, tiny ensembles and shallow
    trees: positions and tree sizes equal the external-callable run."""
    import blackjax_amd as bjx

    rng = np.random.default_rng(3)
    for _ in range(6):
        D = int(rng.integers(3, 128)) * 4
        N = int(rng.choice([1, 3, 70, 300]))
        depth = int(rng.integers(1, 7))
        T = int(rng.integers(1, 6))
        g = torch.Generator(device=dev)
        g.manual_seed(D)
        fn = bjx.targets.NealFunnel()
        q0 = 0.2 * torch.randn(N, D, device=dev, generator=g)
        alg = bjx.nuts(fn, 0.12, torch.ones(D, device=dev), max_num_doublings=depth)
        st0 = alg.init(q0)
        key = bjx.random.key(int(rng.integers(1, 1000)))
        _, pos_a, info_a = alg.run(key, st0, T)
        _, pos_b, info_b = alg.run(key, st0, T, fuse_target=True)
        assert torch.equal(pos_a, pos_b), (N, D, depth, T)
        assert torch.equal(info_a.num_integration_steps, info_b.num_integration_steps)
        assert torch.equal(info_a.acceptance_rate, info_b.acceptance_rate)
