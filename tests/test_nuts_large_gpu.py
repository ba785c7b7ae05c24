"""Larger-size NUTS checks: the HIP-graph driver (device control block, device-side compaction,
fused post+pre) and the eager driver must agree bit for bit, and size-independent tree invariants
must hold at a BASELINE.json-like shape (Neal's funnel, D = 256)."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx

pytestmark = pytest.mark.gpu


def test_graph_and_eager_drivers_agree_at_scale(dev):
    N, D = 4096, 256
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.NealFunnel()
    imm = torch.ones(D, device=dev)
    a_e = bjx.nuts(fn, 0.1, imm, max_num_doublings=8)
    a_g = bjx.nuts(fn, 0.1, imm, max_num_doublings=8, use_graph=True, graph_sync_every=2)
    s_e, s_g = a_e.init(q0), a_g.init(q0)
    for k in bjx.random.split(bjx.random.key(1), 3):
        s_e, i_e = a_e.step(k, s_e)
        s_g, i_g = a_g.step(k, s_g)
        assert torch.equal(s_e.position, s_g.position)
        assert torch.equal(i_e.num_integration_steps, i_g.num_integration_steps)
        assert torch.equal(i_e.num_trajectory_expansions, i_g.num_trajectory_expansions)
        assert torch.equal(i_e.is_turning, i_g.is_turning) and torch.equal(i_e.is_divergent, i_g.is_divergent)
        assert torch.equal(i_e.acceptance_rate, i_g.acceptance_rate)
        assert torch.equal(i_e.trajectory_leftmost_state.position, i_g.trajectory_leftmost_state.position)
        # invariants (nuts.py:286-291, trajectory.py:616-727)
        n = i_e.num_integration_steps
        d = i_e.num_trajectory_expansions
        assert int(n.min()) >= 1 and int(d.min()) >= 1 and int(d.max()) <= 8
        assert bool((n <= (2 ** d.long() - 1)).all())          # at most 2^depth - 1 new states
        assert bool((n > (2 ** (d.long() - 1) - 1)).all())     # the last doubling added >= 1 state
        stopped = i_e.is_turning | i_e.is_divergent
        assert bool((stopped | (d == 8)).all())                 # a chain only stops for a reason
        acc = i_e.acceptance_rate
        assert bool(((acc >= 0) & (acc <= 1 + 1e-5)).all())
        assert torch.isfinite(s_e.position).all() and torch.isfinite(i_e.energy).all()
    assert len(torch.unique(i_e.num_trajectory_expansions)) >= 3  # a real mix of depths
