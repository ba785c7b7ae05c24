"""GPU parity of the dense-metric path (fp32 MFMA GEMMs) vs the oracle.  The MFMA accumulates in
fp32 in a fixed k order while the oracle accumulates in fp64, so this path is compared with a
stated tolerance (RTOL) rather than bit-for-bit; accept/reject decisions must agree except where
the uniform draw is within the energy tolerance of p_accept (counted, must be rare)."""
import json
import os

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import _lib
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
RTOL = 2e-5
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


@pytest.mark.parametrize("N,D", [(300, 512), (7, 6), (129, 130), (1, 4), (256, 128), (33, 17)])
def test_dense_matmul(dev, N, D):
    g = torch.Generator(device=dev)
    g.manual_seed(N * 1000 + D)
    A = torch.randn(N, D, device=dev, generator=g)
    B = torch.randn(D, D, device=dev, generator=g)
    C = torch.full((N, D), float("nan"), device=dev)
    _lib.call("bjx_dense_matmul", _lib.current_stream(), N, D, A.data_ptr(), B.data_ptr(), C.data_ptr())
    ref = (A.double() @ B.double())
    scale = (A.double().abs() @ B.double().abs())
    assert torch.isfinite(C).all()
    assert float(((C.double() - ref).abs() / scale).max()) < 1e-6


def test_reference_golden_velocity_verlet_through_dense_kernels(dev):
    """tests/mcmc/test_integrators.py:74-103,136-145 of the reference: dense 6-d Gaussian, 16 steps."""
    k = KATS["velocity_verlet_mvnormal"]
    cov = torch.tensor(k["cov"], device=dev)
    P = torch.linalg.inv(cov.double())
    q = torch.tensor([k["q_init"]], device=dev)
    p = torch.tensor([k["p_init"]], device=dev)
    grad = lambda x: (-(x.double() @ P.T)).float()
    g = grad(q)
    s = _lib.current_stream()
    eps = k["step_size"]
    for i in range(k["num_steps"]):
        p_new = torch.empty_like(p)
        _lib.call("bjx_leapfrog_dense", s, 1, 6, 1 if i == 0 else 2, eps, None, cov.data_ptr(),
                  q.data_ptr(), p.data_ptr(), g.data_ptr(), q.data_ptr(), p_new.data_ptr())
        p = p_new
        g = grad(q)
    p = p + (eps * 0.5) * g  # closing half kick
    np.testing.assert_allclose(t2n(q)[0], k["q_final"], atol=k["atol"])
    np.testing.assert_allclose(t2n(p)[0], k["p_final"], atol=k["atol"])


@pytest.mark.parametrize("N,D,L", [(100, 64, 8), (37, 30, 5)])
def test_dense_hmc_vs_oracle(dev, N, D, L):
    """Scaled-down configs[4]: AR(1) correlated Gaussian, dense imm = Sigma."""
    rho = 0.9
    fn_o = otargets.ar1_gaussian(rho, D)
    cov = otargets.ar1_covariance(rho, D)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    q0 = prng.normal(prng.key(1), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.hmc(tgt, 0.5, dev_t(cov, dev), L, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev))
    np.testing.assert_array_equal(t2n(st_g.logdensity_grad), st_o.logdensity_grad)
    near_ties = 0
    for kk in prng.split(prng.key(0), 4):
        st_o_new, info_o = ohmc.kernel(kk, st_o, fn_o, np.float32(0.5), cov, L, chain_offset=3)
        st_g, info_g = alg.step(kk, st_g)
        np.testing.assert_allclose(t2n(info_g.momentum), info_o.momentum, rtol=RTOL, atol=RTOL)
        np.testing.assert_allclose(t2n(info_g.proposal.position), info_o.proposal.position, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-3, atol=1e-4)
        acc_g, acc_o = t2n(info_g.is_accepted), info_o.is_accepted
        mism = acc_g != acc_o
        if mism.any():  # only legitimate when u is within the energy tolerance of p_accept
            kc = prng.split(prng.split(kk, N, offset=3), 2)[:, 1]
            u = prng.uniform(kc, ())
            assert np.all(np.abs(u[mism] - info_o.acceptance_rate[mism]) < 1e-3)
            near_ties += int(mism.sum())
        # continue both from the ORACLE's state so a legitimate near-tie flip cannot cascade
        st_o = st_o_new
        st_g = bjx.hmc.init(dev_t(st_o.position, dev), tgt)
        assert 0.3 < info_o.acceptance_rate.mean() <= 1.0
    assert near_ties <= 1


def test_dense_hmc_statistics(dev):
    N, D, L = 2048, 16, 12
    rho = 0.9
    tgt = bjx.targets.AR1Gaussian(rho, D)
    cov = tgt.covariance(dev)
    alg = bjx.hmc(tgt, 0.6, cov, L)
    st = alg.init(torch.randn(N, D, device=dev))
    accs = []
    for kk in bjx.random.split(bjx.random.key(3), 25):
        st, info = alg.step(kk, st)
        accs.append(info.acceptance_rate.mean().item())
    x = st.position.double()
    emp = (x.T @ x) / N
    assert float((emp - cov.double()).abs().max()) < 0.15
    assert np.mean(accs) > 0.7
