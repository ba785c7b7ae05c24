"""The free-running tick of the contract path, k_nuts_async_tick3<NI, W> (csrc/bjx_nuts_tick.hip: lean leaf + deferred
transition ends, one launch per tick; NI = 1 .. 4 sixteen-byte pieces per lane = rows of at most 256 .. 1 024 floats),
at every row width against the LOCKSTEP kernels -- an independent implementation of the same transitions
(bjx_nuts_pre / post / merge) -- and against the oracle.  Round 5 removed the tick variants this file used to
compare with each other (the v2 leaf, four chains per wave, the work-list kernel, and their environment switches);
what is left is one kernel per shape, and every one of them is named here:

  k_nuts_async_tick3<1, 4>  D <= 256      k_nuts_async_tick3<2, 4>  D <= 512
  k_nuts_async_tick3<3, 2>  D <= 768      k_nuts_async_tick3<4, 2>  D <= 1 024
  k_nuts_async_fused<4, 0>  16-byte rows beyond 1 024 floats;  k_nuts_async_fused<1, 0>  D % 4 != 0

Reference: blackjax/mcmc/nuts.py:113-145, trajectory.py:242-395, 580-727."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc, nuts as onuts, prng, targets as otargets

pytestmark = pytest.mark.gpu
FIELDS = ("num_integration_steps", "num_trajectory_expansions", "is_divergent", "is_turning", "energy", "acceptance_rate")


def _run_vs_steps(alg, st0, key, T, key_layout="step_major"):
    final, pos, info = alg.run(key, st0, T, key_layout=key_layout)
    st = st0
    for t in range(T):
        k = prng.split(key, T)[t] if key_layout == "step_major" else bjx.random.ChainMajorKey(key, t)
        st, inf = alg.step(k, st)
        for f in FIELDS:
            assert torch.equal(getattr(info, f)[t], getattr(inf, f)), (t, f)
        assert torch.equal(pos[t], st.position), t
    assert torch.equal(final.position, st.position) and torch.equal(final.logdensity_grad, st.logdensity_grad)
    return pos, info


@pytest.mark.parametrize("D", [8, 64, 100, 128, 200, 256, 260, 384, 512, 640, 768, 1024, 1028, 1540, 30, 258])
def test_free_running_tick_equals_lockstep_at_every_row_width(dev, D):
    N = 301 if D != 256 else 1030
    g = torch.Generator(device=dev).manual_seed(D)
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=8)
    pos, info = _run_vs_steps(alg, alg.init(q0), bjx.random.key(42), 4)
    if D == 256:
        assert len(torch.unique(info.num_trajectory_expansions)) >= 4  # trees of several depths were compared


def test_per_chain_parameters_chain_major_keys_and_the_oracle(dev):
    """Per-chain step sizes and per-chain diagonal metrics, chain-major keys, a chain offset; decisions exact against the
    oracle, positions within the stated 1e-6."""
    N, D, T = 50, 36, 4
    rng = np.random.default_rng(3)
    q0 = (0.5 * prng.normal(prng.key(2), (N, D))).astype(np.float32)
    eps = rng.uniform(0.05, 0.6, N).astype(np.float32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(np.float32)
    alg = bjx.nuts(bjx.targets.NealFunnel(), torch.as_tensor(eps, device=dev),
                   bjx.metrics.PerChainDiag(torch.as_tensor(imm, device=dev)), max_num_doublings=5, chain_offset=11)
    pos, info = _run_vs_steps(alg, alg.init(torch.as_tensor(q0, device=dev)), bjx.random.key(42), T, "chain_major")
    assert int(info.num_trajectory_expansions.max()) == 5
    fn_o = otargets.neal_funnel()
    st_o = ohmc.init(q0, fn_o)
    for t in range(T):
        ck = prng.split(prng.split(prng.key(42), N, offset=11), T)[:, t]
        st_o, info_o = onuts.kernel(None, st_o, fn_o, eps, imm, 5, chain_keys_override=ck, per_chain_diag=True)
        assert np.array_equal(info.num_integration_steps[t].cpu().numpy(), info_o.num_integration_steps), t
        assert np.array_equal(info.is_divergent[t].cpu().numpy(), info_o.is_divergent)
        assert np.array_equal(info.is_turning[t].cpu().numpy(), info_o.is_turning)
        np.testing.assert_allclose(pos[t].cpu().numpy(), st_o.position, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(info.energy[t].cpu().numpy(), info_o.energy, rtol=1e-6, atol=1e-6)


def test_shared_nontrivial_metric_and_a_multi_stage_integrator(dev):
    N, D = 64, 256
    sig = (10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32)
    q0 = torch.as_tensor((prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32), device=dev)
    alg = bjx.nuts(bjx.targets.DiagGaussian(torch.as_tensor(1.0 / (sig * sig), device=dev)), 0.3,
                   torch.as_tensor(sig * sig, device=dev), max_num_doublings=6)
    _run_vs_steps(alg, alg.init(q0), bjx.random.key(42), 4)
    g = torch.Generator(device=dev).manual_seed(0)
    for D2 in (200, 640):  # a leaf lasts three ticks (two middle stages + the closing tick); 640: three pieces per lane
        q0 = 0.1 * torch.randn(70, D2, device=dev, generator=g)
        alg = bjx.nuts(bjx.targets.NealFunnel(), 0.15, torch.ones(D2, device=dev), max_num_doublings=6,
                       integrator=bjx.integrators.yoshida)
        _run_vs_steps(alg, alg.init(q0), bjx.random.key(9), 3)


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_tail_entered_with_one_live_chain_keeps_chain_zero_intact(dev, seed):
    """Regression (round 5): a run whose recorded tail starts with exactly ONE live chain used to build its pending-
    position buffer as ``q[:1].expand(1, D).contiguous()`` -- a VIEW of chain 0's state row -- so that chain's
    leapfrog positions overwrote ``final.position[0]`` (log-density and gradient stayed right).  These three seeds
    reproduced it deterministically (N = 70, D = 1 024, yoshida: a leaf lasts three ticks)."""
    N, D, T = 70, 1024, 3
    tgt = bjx.targets.NealFunnel()
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    alg = bjx.nuts(tgt, 0.15, torch.ones(D, device=dev), max_num_doublings=6, integrator=bjx.integrators.yoshida)
    final, pos, info = alg.run(bjx.random.key(9), alg.init(q0), T)
    assert torch.equal(final.position, pos[-1])
    lp, gr = tgt(final.position)
    assert torch.equal(gr, final.logdensity_grad) and torch.equal(lp, final.logdensity)
