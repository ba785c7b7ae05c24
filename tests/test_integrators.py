"""Higher-order palindromic integrators.

CPU: the oracle's generalized_two_stage_integrator reproduces the reference's golden vector with
every integrator (tests/mcmc/test_integrators.py:136-223 checks all of them against the same
end point at atol 1e-2 and energy drift < its `precision`).  GPU: hmc with mclachlan / yoshida /
omelyan matches the oracle bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hmc as ohmc
from oracle import integrators as oint
from oracle import prng, targets as otargets
from oracle.fp import f32, f64

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


@pytest.mark.parametrize("name", ["velocity_verlet", "mclachlan", "yoshida", "omelyan"])
def test_oracle_integrators_on_reference_golden(name):
    k = KATS["velocity_verlet_mvnormal"]
    cov = np.array(k["cov"], dtype=f32)
    P = np.linalg.inv(cov.astype(f64))

    def fn(q):
        g = -(q.astype(f64) @ P.T)
        return (0.5 * np.sum(q * g, -1)).astype(f32), g.astype(f32)

    q = np.array([k["q_init"]], dtype=f32)
    p = np.array([k["p_init"]], dtype=f32)
    metric = ohmc.default_metric(cov)
    lp, g = fn(q)
    z = ohmc.IntegratorState(q, p, lp, g)
    e0 = ohmc.hmc_energy(metric, z)
    for _ in range(k["num_steps"]):
        z = oint.one_step(z, k["step_size"], fn, metric, getattr(oint, name))
    # the reference asserts atol=1e-2 on the position for every integrator and energy conservation
    np.testing.assert_allclose(z.position[0], k["q_final"], atol=1e-2)
    assert abs(float(ohmc.hmc_energy(metric, z)[0] - e0[0])) < 1e-4
    if name == "velocity_verlet":
        np.testing.assert_allclose(z.position[0], k["q_final"], atol=k["atol"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mclachlan", "yoshida", "omelyan"])
def test_hmc_higher_order_integrators_gpu_parity(dev, name):
    import blackjax_amd as bjx

    N, D, L = 50, 96, 4
    rng = np.random.default_rng(1)
    sig = (10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))).astype(np.float32)
    inv_var = (np.float32(1) / (sig * sig)).astype(np.float32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(np.float32)
    eps = rng.uniform(0.1, 0.5, N).astype(np.float32)
    fn_o = otargets.diag_gaussian(inv_var)
    q0 = (prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    to_dev = lambda a: torch.as_tensor(a, device=dev)
    alg = bjx.hmc(bjx.targets.DiagGaussian(to_dev(inv_var)), to_dev(eps), to_dev(imm), L,
                  integrator=getattr(bjx.integrators, name), chain_block=16)
    st_g = alg.init(to_dev(q0))
    n_rej = 0
    for kk in prng.split(prng.key(0), 4):
        st_o, (p_acc, acc, div, e1, z) = oint.hmc_kernel(kk, st_o, fn_o, eps, imm, L, getattr(oint, name))
        st_g, info = alg.step(kk, st_g)
        assert np.array_equal(info.is_accepted.cpu().numpy(), acc)
        assert np.array_equal(st_g.position.cpu().numpy(), st_o.position)
        assert np.array_equal(info.proposal.position.cpu().numpy(), z.position)
        np.testing.assert_allclose(info.acceptance_rate.cpu().numpy(), p_acc, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(info.energy.cpu().numpy(), e1, rtol=1e-6)
        n_rej += int((~acc).sum())
    assert n_rej >= 0
    # higher-order integrators conserve energy better than velocity Verlet at equal step size
    vv = bjx.hmc(bjx.targets.DiagGaussian(to_dev(inv_var)), 0.4, to_dev(sig * sig), L)
    ho = bjx.hmc(bjx.targets.DiagGaussian(to_dev(inv_var)), 0.4, to_dev(sig * sig), L,
                 integrator=getattr(bjx.integrators, name))
    s0 = vv.init(to_dev(q0))
    _, i_vv = vv.step(bjx.random.key(3), s0)
    _, i_ho = ho.step(bjx.random.key(3), s0)
    assert float(i_ho.acceptance_rate.mean()) >= float(i_vv.acceptance_rate.mean()) - 0.02


def test_unsupported_integrator_combinations():
    import blackjax_amd as bjx

    # round 3: every sampler takes the palindromic integrators (tests/test_integrators_samplers_gpu.py);
    # what is not an Integrator is still refused
    bjx.nuts.build_kernel(bjx.integrators.mclachlan)
    with pytest.raises(NotImplementedError):
        bjx.nuts.build_kernel(object())
    with pytest.raises(NotImplementedError):
        bjx.hmc.build_kernel(object())
