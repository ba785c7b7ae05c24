"""Seeded random-shape parity sweep: HIP path vs oracle for hmc / ghmc / nuts over shapes, chain blocks and
parameter forms nobody picked by hand (ragged N, D not a multiple of 4, D on either side of the
short-row / fused-first / register-resident thresholds, per-chain vs shared parameters).  Accept bits,
divergence flags and tree sizes exact; positions within 1e-6."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import ghmc as oghmc
from oracle import hmc as ohmc
from oracle import nuts as onuts
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def _cases(seed, n, d_choices):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        D = int(rng.choice(d_choices))
        N = int(rng.integers(1, 70))
        out.append((i, N, D, int(rng.integers(1, 7)), bool(rng.integers(0, 2)), int(rng.choice([0, 7, 16, 33]))))
    return out


@pytest.mark.parametrize("i,N,D,L,per_chain,chain_block", _cases(11, 20, [1, 3, 8, 37, 64, 100, 128, 129, 132, 260, 300, 512, 1028]))
def test_hmc_random_shapes(dev, i, N, D, L, per_chain, chain_block):
    rng = np.random.default_rng(1000 + i)
    sig = (10.0 ** rng.uniform(-0.5, 0.5, D)).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(i), (N, D)) * sig).astype(f32)
    eps = rng.uniform(0.05, 0.5, N).astype(f32) if per_chain else f32(rng.uniform(0.05, 0.5))
    imm = (sig * sig * rng.uniform(0.5, 2.0, (N, D) if per_chain else D)).astype(f32)
    fn_o = otargets.diag_gaussian(inv_var)
    imm_g = bjx.metrics.PerChainDiag(dev_t(imm, dev)) if per_chain else dev_t(imm, dev)
    alg = bjx.hmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), dev_t(eps, dev) if per_chain else float(eps), imm_g, L,
                  chain_offset=5, chain_block=chain_block or None, use_graph=False)
    st_g, st_o = alg.init(dev_t(q0, dev)), ohmc.init(q0, fn_o)
    for k in prng.split(prng.key(77 + i), 3):
        st_o, info_o = ohmc.kernel(k, st_o, fn_o, eps, imm, L, chain_offset=5, per_chain_diag=per_chain)
        st_g, info_g = alg.step(k, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(info_g.proposal.momentum), info_o.proposal.momentum, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("i,N,D,L,per_chain,chain_block", _cases(12, 10, [1, 6, 63, 64, 130, 256, 257, 516, 1024, 1028]))
def test_ghmc_random_shapes(dev, i, N, D, L, per_chain, chain_block):
    del L, chain_block
    rng = np.random.default_rng(2000 + i)
    sig = (10.0 ** rng.uniform(-0.5, 0.5, D)).astype(f32)
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (prng.normal(prng.key(i), (N, D)) * sig).astype(f32)
    if per_chain:
        eps, alpha, delta = (rng.uniform(0.1, 0.9, N).astype(f32), rng.uniform(0.05, 0.95, N).astype(f32),
                             rng.uniform(0.0, 0.5, N).astype(f32))
        scale = (sig * rng.uniform(0.7, 1.4, (N, D))).astype(f32)
        args_g = (dev_t(eps, dev), dev_t(scale, dev), dev_t(alpha, dev), dev_t(delta, dev))
    else:
        eps, alpha, delta, scale = f32(0.6), f32(0.3), f32(0.15), sig
        args_g = (0.6, dev_t(scale, dev), 0.3, 0.15)
    fn_o = otargets.diag_gaussian(inv_var)
    alg = bjx.ghmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), *args_g)
    st_g, st_o = alg.init(dev_t(q0, dev), prng.key(i + 9)), oghmc.init(q0, fn_o, prng.key(i + 9))
    for k in prng.split(prng.key(55 + i), 4):
        st_o, info_o = oghmc.kernel(k, st_o, fn_o, eps, scale, alpha, delta)
        st_g, info_g = alg.step(k, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(st_g.momentum), st_o.momentum, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(st_g.slice), st_o.slice, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("i,N,D", [(0, 9, 3), (1, 17, 12), (2, 5, 20), (3, 12, 8), (4, 3, 33)])
def test_nuts_random_shapes_lockstep_and_free_running(dev, i, N, D):
    rng = np.random.default_rng(3000 + i)
    q0 = (0.7 * prng.normal(prng.key(i), (N, D))).astype(f32)
    eps = rng.uniform(0.1, 0.5, N).astype(f32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(f32)
    fn_o = otargets.neal_funnel()
    alg = bjx.nuts(bjx.targets.NealFunnel(), dev_t(eps, dev), bjx.metrics.PerChainDiag(dev_t(imm, dev)),
                   max_num_doublings=5)
    st0 = alg.init(dev_t(q0, dev))
    T = 3
    final, positions, info = alg.run(prng.key(40 + i), st0, T)
    st_g, st_o = st0, ohmc.init(q0, fn_o)
    for t, k in enumerate(prng.split(prng.key(40 + i), T)):
        st_o, info_o = onuts.kernel(k, st_o, fn_o, eps, imm, 5, per_chain_diag=True)
        st_g, info_g = alg.step(k, st_g)
        assert np.array_equal(t2n(info_g.num_integration_steps), info_o.num_integration_steps)
        assert np.array_equal(t2n(info.num_integration_steps[t]), info_o.num_integration_steps)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-6, atol=1e-6)
        assert torch.equal(positions[t], st_g.position)
