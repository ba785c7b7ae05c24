"""window_adaptation(nuts) with free-running chains == the lockstep warm-up, bit for bit: chains are
independent in the reference's vmapped warm-up (adaptation/staged_adaptation.py:186-249, 860-876), so
letting every chain adapt when IT finishes a transition must not change anything."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from oracle import prng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("N,D,T,max_depth,shrink", [(48, 8, 130, 5, 0.0), (33, 256, 40, 4, 0.0),
                                                     (16, 12, 19, 3, 0.0), (10, 7, 130, 4, 0.0),
                                                     (21, 20, 150, 4, 2.5)])
def test_free_running_warmup_equals_lockstep(dev, N, D, T, max_depth, shrink):
    g = torch.Generator(device=dev)
    g.manual_seed(N + D)
    inv_var = (torch.rand(D, device=dev, generator=g) * 3.0 + 0.2).contiguous()
    fn = bjx.targets.DiagGaussian(inv_var)
    q0 = torch.randn(N, D, device=dev, generator=g)
    kw = dict(initial_step_size=0.7, target_acceptance_rate=0.8, max_num_doublings=max_depth,
              imm_shrinkage_to_previous=shrink)
    warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn(
        set(), {"acceptance_rate", "num_integration_steps"}, {"step_size"}), **kw)
    key = prng.key(11)
    (st_l, par_l), hist = warm.run(key, q0, T, chain_offset=5)
    with pytest.raises(ValueError):  # a custom adaptation_info_fn is not silently ignored
        warm.run(key, q0, T, chain_offset=5, free_running=True)
    warm_f = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, **kw)
    (st_f, par_f), info = warm_f.run(key, q0, T, chain_offset=5, free_running=True)
    assert torch.equal(st_f.position, st_l.position)
    assert torch.equal(st_f.logdensity, st_l.logdensity)
    assert torch.equal(st_f.logdensity_grad, st_l.logdensity_grad)
    assert torch.equal(par_f["step_size"], par_l["step_size"])
    assert torch.equal(par_f["inverse_mass_matrix"], par_l["inverse_mass_matrix"])
    # per-step records
    assert torch.equal(info.acceptance_rate, hist.info.acceptance_rate)
    assert torch.equal(info.num_integration_steps.to(hist.info.num_integration_steps.dtype),
                       hist.info.num_integration_steps)
    assert torch.equal(info.step_size, hist.adaptation_state.step_size)
    assert np.isfinite(par_f["step_size"].cpu().numpy()).all()


def test_free_running_warmup_funnel_per_chain_depths(dev):
    """Neal's funnel: tree depths differ wildly between chains, so chains reach their window ends at
    very different ticks."""
    N, D, T = 64, 16, 60
    fn = bjx.targets.NealFunnel()
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    q0 = 0.5 * torch.randn(N, D, device=dev, generator=g)
    warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, initial_step_size=0.3,
                                 max_num_doublings=6)
    (st_l, par_l), _ = warm.run(prng.key(2), q0, T)
    (st_f, par_f), info = warm.run(prng.key(2), q0, T, free_running=True)
    assert torch.equal(st_f.position, st_l.position)
    assert torch.equal(par_f["step_size"], par_l["step_size"])
    assert torch.equal(par_f["inverse_mass_matrix"], par_l["inverse_mass_matrix"])
    assert int(info.num_trajectory_expansions.max()) >= 4


@pytest.mark.parametrize("N,D,T", [(96, 64, 60), (4500, 256, 12)])
def test_free_running_warmup_with_an_engine_resident_target(dev, N, D, T):
    """``run(..., free_running=True, fuse_target=True)``: the tick kernels evaluate the funnel themselves, many
    ticks per launch, with the per-chain dual-averaging / Welford updates at each chain's own transition ends --
    final state, step sizes, metrics and every record equal those of the external-callable run bit for bit."""
    fn = bjx.targets.NealFunnel()
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    q0 = 0.5 * torch.randn(N, D, device=dev, generator=g)
    warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, initial_step_size=0.3,
                                 max_num_doublings=6)
    (st_a, par_a), info_a = warm.run(prng.key(2), q0, T, free_running=True)
    (st_b, par_b), info_b = warm.run(prng.key(2), q0, T, free_running=True, fuse_target=True)
    for a, b in zip(st_a, st_b):
        assert torch.equal(a, b)
    assert torch.equal(par_a["step_size"], par_b["step_size"])
    assert torch.equal(par_a["inverse_mass_matrix"], par_b["inverse_mass_matrix"])
    for name in ("logdensity", "acceptance_rate", "energy", "num_integration_steps", "num_trajectory_expansions",
                 "is_divergent", "is_turning", "step_size"):
        assert torch.equal(getattr(info_a, name), getattr(info_b, name)), name
    with pytest.raises(ValueError):
        warm.run(prng.key(2), q0, T, fuse_target=True)


@pytest.mark.parametrize("name", ["mclachlan", "omelyan"])
def test_free_running_warmup_with_a_multi_stage_integrator_equals_lockstep(dev, name):
    """Round 4: the free-running warm-up takes any palindromic integrator the free-running tick kernels take
    (a leaf lasts K ticks); step sizes, metrics, final states and per-step records equal the lockstep warm-up
    with the same integrator bit for bit."""
    N, D, T = 40, 64, 60
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    inv_var = (torch.rand(D, device=dev, generator=g) * 3.0 + 0.2).contiguous()
    fn = bjx.targets.DiagGaussian(inv_var)
    q0 = torch.randn(N, D, device=dev, generator=g)
    integ = getattr(bjx.integrators, name)
    kw = dict(initial_step_size=0.7, max_num_doublings=5, integrator=integ)
    warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn(
        set(), {"acceptance_rate", "num_integration_steps"}, {"step_size"}), **kw)
    (st_l, par_l), hist = warm.run(prng.key(5), q0, T, chain_offset=2)
    warm_f = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, **kw)
    (st_f, par_f), info = warm_f.run(prng.key(5), q0, T, chain_offset=2, free_running=True)
    assert torch.equal(st_f.position, st_l.position) and torch.equal(st_f.logdensity_grad, st_l.logdensity_grad)
    assert torch.equal(par_f["step_size"], par_l["step_size"])
    assert torch.equal(par_f["inverse_mass_matrix"], par_l["inverse_mass_matrix"])
    assert torch.equal(info.acceptance_rate, hist.info.acceptance_rate)
    assert torch.equal(info.num_integration_steps.to(hist.info.num_integration_steps.dtype),
                       hist.info.num_integration_steps)
