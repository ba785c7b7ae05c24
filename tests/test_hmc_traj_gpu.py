"""``hmc(..., fuse_target=True)``: a whole HMC transition per launch for the log-densities the engine
evaluates itself (``bjx_hmc_trajectory_diag``, csrc/bjx_traj.hip).  Outside the external-callable contract
(the reference calls ``logdensity_fn`` between two leapfrogs, blackjax/mcmc/hmc.py:279-312), so the bar is:
bit for bit the state and info of the default path -- which tests/test_hmc_gpu.py holds against the oracle."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx

pytestmark = pytest.mark.gpu


def _case(dev, target, N, D, per_chain):
    g = torch.Generator(device=dev)
    g.manual_seed(N + D)
    if target == "gauss":
        sig = (10.0 ** (-0.5 + 1.0 * torch.arange(D, device=dev) / (D - 1))).float()
        fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
        q0 = sig * torch.randn(N, D, device=dev, generator=g)
        imm = (sig * sig).contiguous()
        eps = 0.3
    else:
        fn = bjx.targets.NealFunnel()
        q0 = 0.3 * torch.randn(N, D, device=dev, generator=g)
        imm = torch.ones(D, device=dev)
        eps = 0.12
    if per_chain:
        imm = bjx.metrics.PerChainDiag((imm * (0.5 + 1.5 * torch.rand(N, D, device=dev, generator=g))).contiguous())
        eps = (eps * (0.5 + torch.rand(N, device=dev, generator=g))).contiguous()
    return fn, q0, imm, eps


@pytest.mark.parametrize("target,N,D,L,per_chain", [("gauss", 300, 1024, 7, False), ("gauss", 77, 256, 1, True),
                                                    ("gauss", 64, 320, 4, False), ("funnel", 500, 256, 6, False),
                                                    ("funnel", 33, 1024, 5, True), ("funnel", 40, 132, 3, False)])
def test_fused_trajectory_equals_the_default_path(dev, target, N, D, L, per_chain):
    fn, q0, imm, eps = _case(dev, target, N, D, per_chain)
    ref = bjx.hmc(fn, eps, imm, L, chain_offset=3)
    fused = bjx.hmc(fn, eps, imm, L, chain_offset=3, fuse_target=True)
    sa = sb = ref.init(q0)
    n_rej = 0
    for key in [bjx.random.key(1), bjx.random.key(2), bjx.random.ChainMajorKey(bjx.random.key(5), 4)]:
        sa, ia = ref.step(key, sa)
        sb, ib = fused.step(key, sb)
        for a, b in zip(sa, sb):
            assert torch.equal(a, b)
        for name in ("momentum", "acceptance_rate", "is_accepted", "is_divergent", "energy"):
            assert torch.equal(getattr(ia, name), getattr(ib, name)), name
        for a, b in zip(ia.proposal, ib.proposal):
            assert torch.equal(a, b)
        assert ib.num_integration_steps == L
        n_rej += int((~ia.is_accepted).sum())
    if target == "funnel" and N >= 300:
        assert n_rej > 0  # both branches of the select ran
    lean = bjx.hmc(fn, eps, imm, L, chain_offset=3, fuse_target="lean")
    sc, ic = lean.step(bjx.random.key(9), sa)
    sd, _ = ref.step(bjx.random.key(9), sa)
    assert torch.equal(sc.position, sd.position) and ic.momentum is None and ic.proposal is None


def test_fused_trajectory_is_refused_where_it_does_not_apply(dev):
    fn = bjx.targets.DiagGaussian(torch.ones(64, device=dev))
    alg = bjx.hmc(fn, 0.1, torch.ones(64, device=dev), 3, fuse_target=True)
    with pytest.raises(NotImplementedError):  # rows of at most 128 floats reduce in another order
        alg.step(bjx.random.key(0), alg.init(torch.zeros(4, 64, device=dev)))
    plain = lambda q: -0.5 * (q * q).sum(-1)  # noqa: E731  (not a library target)
    alg = bjx.hmc(plain, 0.1, torch.ones(256, device=dev), 3, fuse_target=True)
    with pytest.raises(NotImplementedError):
        alg.step(bjx.random.key(0), alg.init(torch.zeros(4, 256, device=dev)))
    with pytest.raises(NotImplementedError):
        bjx.hmc(fn, 0.1, torch.ones(64, device=dev), 3, fuse_target=True, integrator=bjx.integrators.mclachlan)
    dense = bjx.hmc(bjx.targets.NealFunnel(), 0.1, torch.eye(256, device=dev), 3, fuse_target=True)
    with pytest.raises(NotImplementedError):
        dense.step(bjx.random.key(0), dense.init(torch.zeros(4, 256, device=dev)))


def test_fused_trajectory_random_shapes(dev):
    """Row lengths that leave lanes / pieces partly empty (D = 132 ... 1 020, not multiples of 256), single
    chains, long trajectories: still the bits of the default path."""
    rng = np.random.default_rng(7)
    for _ in range(8):
        D = int(rng.integers(33, 256)) * 4 + 4
        N = int(rng.choice([1, 2, 5, 63, 130]))
        L = int(rng.integers(1, 12))
        target = "gauss" if rng.random() < 0.5 else "funnel"
        fn, q0, imm, eps = _case(dev, target, N, D, bool(rng.random() < 0.5))
        ref = bjx.hmc(fn, eps, imm, L, chain_offset=11)
        fused = bjx.hmc(fn, eps, imm, L, chain_offset=11, fuse_target=True)
        st = ref.init(q0)
        key = bjx.random.key(int(rng.integers(1, 1000)))
        sa, ia = ref.step(key, st)
        sb, ib = fused.step(key, st)
        for a, b in zip(sa, sb):
            assert torch.equal(a, b), (target, N, D, L)
        assert torch.equal(ia.acceptance_rate, ib.acceptance_rate) and torch.equal(ia.energy, ib.energy)
        assert torch.equal(ia.proposal.momentum, ib.proposal.momentum)


def test_window_adaptation_hmc_with_an_engine_resident_target(dev):
    """window_adaptation(hmc, fuse_target=True): every warm-up transition is one launch; the adapted step sizes,
    metrics and the final state equal the default warm-up's bit for bit."""
    N, D, T = 200, 256, 70
    fn, q0, _, _ = _case(dev, "gauss", N, D, False)
    kw = dict(adaptation_info_fn=None, initial_step_size=0.2, num_integration_steps=6)
    (st_a, par_a), _ = bjx.window_adaptation(bjx.hmc, fn, **kw).run(bjx.random.key(4), q0, T)
    (st_b, par_b), _ = bjx.window_adaptation(bjx.hmc, fn, fuse_target=True, **kw).run(bjx.random.key(4), q0, T)
    assert torch.equal(st_a.position, st_b.position)
    assert torch.equal(par_a["step_size"], par_b["step_size"])
    assert torch.equal(par_a["inverse_mass_matrix"], par_b["inverse_mass_matrix"])
    with pytest.raises(NotImplementedError):
        bjx.window_adaptation(bjx.nuts, fn, fuse_target=True)
