"""``nuts(...).step`` served by ONE free-running transition on a persistent workspace (blackjax_amd/nuts.py,
``step_driver``): the same draws and the same NUTSInfo -- every field, the trajectory ends included -- as the
lockstep tree (blackjax/mcmc/nuts.py:113-145), call after call on the same workspace."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx

pytestmark = pytest.mark.gpu


def _same_info(a, b):
    for name in ("momentum", "is_divergent", "is_turning", "energy", "num_trajectory_expansions",
                 "num_integration_steps", "acceptance_rate"):
        x, y = getattr(a, name), getattr(b, name)
        if x.is_floating_point():
            assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), name
        else:
            assert torch.equal(x, y), name
    for end in ("trajectory_leftmost_state", "trajectory_rightmost_state"):
        ea, eb = getattr(a, end), getattr(b, end)
        for f in ("position", "momentum", "logdensity", "logdensity_grad"):
            assert torch.equal(getattr(ea, f), getattr(eb, f)), (end, f)


def _same_state(a, b):
    assert torch.equal(a.position, b.position)
    assert torch.equal(a.logdensity, b.logdensity)
    assert torch.equal(a.logdensity_grad, b.logdensity_grad)


@pytest.mark.parametrize("N,D,max_depth,eps", [
    (300, 64, 8, 0.1),     # busy phase -> speculative tail
    (4096, 32, 7, 0.15),   # busy phase -> one-stream tail -> speculative tail
    (17, 256, 9, 0.05),    # speculative tail from the first chunk on
    (6, 772, 6, 0.1),      # three pieces per lane
    (2, 1024, 6, 0.1),     # four pieces per lane
])
def test_free_step_equals_lockstep_step(dev, N, D, max_depth, eps):
    g = torch.Generator(device=dev)
    g.manual_seed(N + D)
    q0 = 0.2 * torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.NealFunnel()
    imm = torch.ones(D, device=dev)
    lock = bjx.nuts(fn, eps, imm, max_num_doublings=max_depth, step_driver="lockstep")
    free = bjx.nuts(fn, eps, imm, max_num_doublings=max_depth, step_driver="free")
    st_l = st_f = lock.init(q0)
    for k in bjx.random.split(bjx.random.key(3), 5):  # five calls on ONE workspace
        st_l, info_l = lock.step(k, st_l)
        st_f, info_f = free.step(k, st_f)
        _same_state(st_l, st_f)
        _same_info(info_l, info_f)
    assert int(info_l.num_trajectory_expansions.max()) >= 2


def test_free_step_per_call_step_size_and_metric(dev):
    """The workspace copies step size and metric per call: scalar -> per-chain step size, changing metric values."""
    N, D = 200, 48
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    q0 = 0.3 * torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.NealFunnel()
    keys = bjx.random.split(bjx.random.key(11), 4)
    st_l = st_f = bjx.nuts(fn, 0.1, torch.ones(D, device=dev)).init(q0)
    from blackjax_amd import _nuts as bnuts

    kl = bnuts.build_kernel(use_graph="auto")
    algs = {}
    for i, k in enumerate(keys):
        eps = 0.05 + 0.02 * i if i % 2 == 0 else 0.05 + 0.1 * torch.rand(N, device=dev, generator=g)
        imm = 0.5 + torch.rand(D, device=dev, generator=g)
        st_l, info_l = kl(k, st_l, fn, eps, imm, 7)
        # one top-level object per (eps, imm) pair, as a user adapting between calls would build them; the free
        # driver's workspace is per object, so also drive ONE object through changing arguments via its closure
        alg = bjx.nuts(fn, eps, imm, max_num_doublings=7, step_driver="free")
        algs[i] = alg
        st_f, info_f = alg.step(k, st_f)
        _same_state(st_l, st_f)
        _same_info(info_l, info_f)


def test_free_step_workspace_reuse_with_run(dev):
    """step (persistent workspace) interleaved with run (own buffers): neither disturbs the other."""
    N, D, T = 64, 16, 6
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    q0 = 0.3 * torch.randn(N, D, device=dev, generator=g)
    alg = bjx.nuts(bjx.targets.NealFunnel(), 0.2, torch.ones(D, device=dev), max_num_doublings=6)
    ref = bjx.nuts(bjx.targets.NealFunnel(), 0.2, torch.ones(D, device=dev), max_num_doublings=6, step_driver="lockstep")
    st0 = alg.init(q0)
    keys = bjx.random.split(bjx.random.key(21), T)
    st_a, st_r = st0, st0
    for t in range(T):
        st_a, ia = alg.step(keys[t], st_a)
        if t == 2:
            alg.run(bjx.random.key(99), st0, 3)  # a whole free-running run in between
        st_r, ir = ref.step(keys[t], st_r)
        _same_state(st_a, st_r)
        _same_info(ia, ir)
    final, positions, info = alg.run(bjx.random.key(21), st0, T)  # run == the step loop (step-major keys)
    assert torch.equal(final.position, st_a.position)
    assert torch.equal(info.num_integration_steps[-1], ia.num_integration_steps)


def test_free_step_shapes_come_and_go(dev):
    """Three ensemble sizes through ONE algorithm object (the driver keeps the two most recent workspaces), shallow
    depth limits, and the lockstep fall-back for a ChainMajorKey."""
    fn = bjx.targets.NealFunnel()
    D = 32
    imm = torch.ones(D, device=dev)
    for max_depth in (1, 2, 5):
        free = bjx.nuts(fn, 0.2, imm, max_num_doublings=max_depth, step_driver="auto")
        lock = bjx.nuts(fn, 0.2, imm, max_num_doublings=max_depth, step_driver="lockstep")
        for N in (8, 40, 8, 130):
            g = torch.Generator(device=dev)
            g.manual_seed(N)
            st_f = st_l = lock.init(0.3 * torch.randn(N, D, device=dev, generator=g))
            for k in bjx.random.split(bjx.random.key(N), 3):
                st_l, il = lock.step(k, st_l)
                st_f, if_ = free.step(k, st_f)
                _same_state(st_l, st_f)
                _same_info(il, if_)
        ck = bjx.random.ChainMajorKey(bjx.random.key(5), 3)  # (run key, transition): the lockstep tree driver's case
        a, ia = free.step(ck, st_f)
        b, ib = lock.step(ck, st_l)
        _same_state(a, b)
        _same_info(ia, ib)
