"""Free-running NUTS chains with ONE shared dense inverse mass matrix, every product v = M^{-1} p of a tick as
one fp32 MFMA GEMM over the live rows (bjx_nuts_async_t.gemm_*; VERDICT r3 "next" #6,
/root/reference/blackjax/mcmc/nuts.py:150-158, metrics.py:263-304): the ticks are built from the per-chain
device functions of the lockstep kernels with the GEMM supplying the velocities, so ``run(T)`` must equal ``T``
lockstep steps on the GEMM path BIT FOR BIT (those steps are pinned against the oracle's f32-chain mode in
tests/test_nuts_gpu.py::test_nuts_shared_dense_metric_on_the_gemm) -- and, directly, the oracle."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import hmc as ohmc
from oracle import nuts as onuts
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def _steps(alg, key, st, T):
    pos, infos = [], []
    for k in prng.split(key, T):
        st, inf = alg.step(k, st)
        pos.append(st.position.clone())
        infos.append(inf)
    return st, torch.stack(pos), infos


def _same(positions, info, final, st, pos_s, infos):
    T = positions.shape[0]
    for t in range(T):
        assert torch.equal(info.num_integration_steps[t], infos[t].num_integration_steps), t
        assert torch.equal(info.num_trajectory_expansions[t], infos[t].num_trajectory_expansions), t
        assert torch.equal(info.is_turning[t], infos[t].is_turning), t
        assert torch.equal(info.is_divergent[t], infos[t].is_divergent), t
        assert torch.equal(positions[t], pos_s[t]), t
        assert torch.equal(info.energy[t], infos[t].energy), t
        assert torch.equal(info.acceptance_rate[t], infos[t].acceptance_rate), t
    assert torch.equal(final.position, st.position)
    assert torch.equal(final.logdensity, st.logdensity)
    assert torch.equal(final.logdensity_grad, st.logdensity_grad)


@pytest.mark.parametrize("N,D,cap,graph", [(40, 128, 0, "auto"), (600, 128, 0, "auto"), (300, 72, 128, False),
                                           (96, 64, 0, True), (5000, 128, 256, "auto")])
def test_free_running_gemm_equals_lockstep_gemm_steps(dev, monkeypatch, N, D, cap, graph):
    """Same keys, same arithmetic: positions, energies, acceptance rates, tree shapes and flags of every
    transition bit for bit; with a momentum list smaller than the ensemble (chains wait for a slot) and under
    the recorded-sequence drivers (the tail of the run, `run_use_graph=True`)."""
    rho, T = 0.8, 4
    imm = dev_t(otargets.ar1_covariance(rho, D), dev)
    q0 = dev_t(prng.normal(prng.key(6), (N, D)).astype(np.float32), dev)
    alg = bjx.nuts(bjx.targets.AR1Gaussian(rho, D), 0.4, imm, max_num_doublings=5, dense_gemm=True,
                   run_use_graph=graph)
    st0 = alg.init(q0)
    st, pos_s, infos = _steps(alg, prng.key(3), st0, T)
    monkeypatch.setenv("BJX_NUTS_FREE_GEMM", "1")
    if cap:
        import importlib

        monkeypatch.setattr(importlib.import_module("blackjax_amd.nuts"), "_DENSE_GEMM_CAP", cap)
    final, positions, info = alg.run(prng.key(3), st0, T)
    _same(positions, info, final, st, pos_s, infos)
    assert len(set(t2n(info.num_trajectory_expansions).ravel().tolist())) > 1  # chains do leave lockstep


def test_free_running_gemm_funnel_with_divergences(dev, monkeypatch):
    """Neal's funnel under a dense metric and a large step: divergent and turning subtrees, trees of every depth."""
    N, D, T = 256, 64, 3
    rng = np.random.default_rng(0)
    a = rng.standard_normal((D, D)).astype(np.float32) * 0.1
    imm = dev_t((a @ a.T + np.eye(D, dtype=np.float32)).astype(np.float32), dev)
    q0 = dev_t(prng.normal(prng.key(9), (N, D)).astype(np.float32), dev)
    alg = bjx.nuts(bjx.targets.NealFunnel(), 0.6, imm, max_num_doublings=6, dense_gemm=True)
    st0 = alg.init(q0)
    st, pos_s, infos = _steps(alg, prng.key(11), st0, T)
    monkeypatch.setenv("BJX_NUTS_FREE_GEMM", "1")
    final, positions, info = alg.run(prng.key(11), st0, T)
    _same(positions, info, final, st, pos_s, infos)
    assert bool(info.is_divergent.any()) and bool(info.is_turning.any())


def test_free_running_gemm_against_the_oracle(dev):
    """Directly against the oracle's f32-chain mode (the GEMM's stated k order) with the engine's fp32 Cholesky
    factor: one transition per chain through ``run_free(key_layout="step")``."""
    import importlib

    nuts_mod = importlib.import_module("blackjax_amd.nuts")

    N, D, rho = 128, 128, 0.8
    fn_o = otargets.ar1_gaussian(rho, D)
    imm = otargets.ar1_covariance(rho, D)
    q0 = prng.normal(prng.key(6), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    alg = bjx.nuts(tgt, 0.4, dev_t(imm, dev), max_num_doublings=5, dense_gemm=True)
    st_g = alg.init(dev_t(q0, dev))
    m = bjx.metrics.default_metric(dev_t(imm, dev), N, D, dev)
    metric = ohmc.default_metric(imm, dense_accum="f32chain",
                                 mass_matrix_sqrt=np.ascontiguousarray(t2n(m.mass_sqrt_t).T))
    k = prng.split(prng.key(8), 1)[0]
    st_o, info_o = onuts.kernel(k, st_o, fn_o, np.float32(0.4), imm, 5, metric=metric)
    final, positions, info = nuts_mod.run_free(k, st_g, tgt, 0.4, dev_t(imm, dev), 1, 5, key_layout="step",
                                               dense_gemm=True)
    assert np.array_equal(t2n(info.num_integration_steps[0]), info_o.num_integration_steps)
    assert np.array_equal(t2n(info.num_trajectory_expansions[0]), info_o.num_trajectory_expansions)
    assert np.array_equal(t2n(info.is_turning[0]), info_o.is_turning)
    assert np.array_equal(t2n(info.is_divergent[0]), info_o.is_divergent)
    np.testing.assert_allclose(t2n(final.position), st_o.position, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(t2n(info.acceptance_rate[0]), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(t2n(info.energy[0]), info_o.energy, rtol=1e-6, atol=1e-6)


def test_gemm_mode_refuses_what_it_does_not_implement(dev):
    import importlib

    nuts_mod = importlib.import_module("blackjax_amd.nuts")

    N, D = 8, 64
    q0 = torch.zeros(N, D, device=dev)
    tgt = bjx.targets.AR1Gaussian(0.5, D)
    st = bjx.nuts(tgt, 0.1, torch.ones(D, device=dev)).init(q0)
    with pytest.raises(NotImplementedError):  # diagonal metric
        nuts_mod.run_free(prng.key(0), st, tgt, 0.1, torch.ones(D, device=dev), 1, 3, dense_gemm=True)
    with pytest.raises(NotImplementedError):  # multi-stage integrator
        nuts_mod.run_free(prng.key(0), st, tgt, 0.1, torch.eye(D, device=dev), 1, 3, dense_gemm=True,
                          integrator=bjx.integrators.mclachlan)


@pytest.mark.parametrize("N,D", [(256, 128), (4096, 512), (40, 128), (300, 72)])
def test_matmul_bt_equals_matmul(dev, N, D):
    """bjx_dense_matmul_bt (the matrix also given transposed -> the kernel that reads it as stored, for whole aligned
    tiles) returns bjx_dense_matmul's product bit for bit, for a NON-symmetric (triangular) matrix."""
    from blackjax_amd import _lib

    g = torch.Generator(device=dev)
    g.manual_seed(1)
    a = torch.randn(N, D, device=dev, generator=g)
    b = torch.tril(torch.randn(D, D, device=dev, generator=g)).contiguous()
    bt = b.t().contiguous()
    c1, c2 = torch.empty_like(a), torch.empty_like(a)
    s = _lib.current_stream()
    _lib.call("bjx_dense_matmul", s, N, D, a.data_ptr(), b.data_ptr(), c1.data_ptr())
    _lib.call("bjx_dense_matmul_bt", s, N, D, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c2.data_ptr())
    assert torch.equal(c1, c2)
    np.testing.assert_allclose(t2n(c1), t2n(a.double() @ b.double()), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("N,D", [(1, 128), (7, 72), (32, 256), (130, 128), (1000, 256), (4096, 512), (16384, 128),
                                 (300, 72), (5000, 200)])
def test_apply_imm_t_is_the_same_chain_on_every_kernel(dev, N, D):
    """bjx_dense_apply_imm_t: V = P imm^T (the matrix read as stored, metrics.py:263-304) for a matrix that is NOT
    bitwise symmetric.  Few rows run on the latency-oriented kernel, whole aligned tiles on the MFMA kernel that reads
    the matrix as stored, ragged shapes on the general MFMA kernel over the transposed copy: all three are one
    ascending-k fp32 fma chain per element in the MFMA k order -- compared here with bjx_dense_matmul(P, imm_t)
    (always the general MFMA kernel) bit for bit, and with the oracle's restatement of that chain."""
    from blackjax_amd import _lib
    from oracle import fp

    g = torch.Generator(device=dev)
    g.manual_seed(2)
    p = torch.randn(N, D, device=dev, generator=g)
    imm = torch.randn(D, D, device=dev, generator=g).contiguous()  # no symmetry at all
    imm_t = imm.t().contiguous()
    v1, v2 = torch.empty_like(p), torch.empty_like(p)
    s = _lib.current_stream()
    _lib.call("bjx_dense_apply_imm_t", s, N, D, p.data_ptr(), imm.data_ptr(), imm_t.data_ptr(), v1.data_ptr())
    _lib.call("bjx_dense_matmul", s, N, D, p.data_ptr(), imm_t.data_ptr(), v2.data_ptr())
    assert torch.equal(v1, v2)
    rows = slice(0, min(N, 64))
    ref = fp.gemm_f32chain(t2n(p[rows]), t2n(imm_t))
    assert np.array_equal(t2n(v1[rows]), ref)
