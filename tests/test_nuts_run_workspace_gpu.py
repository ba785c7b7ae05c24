"""``nuts(...).run`` on a persistent workspace (blackjax_amd/nuts.py, ``_persistent_run``): buffers and recorded tail
sequences are built once per shape for a CAPACITY of transitions; later runs of any length up to it start the chains'
transition counters at ``capacity - T``.  Every run must equal the per-call driver (``BJX_NUTS_RUN_WS=0``) bit for bit,
and ``T`` calls of ``step`` (blackjax/mcmc/nuts.py:113-145 under ``run_inference_algorithm``'s key layout,
blackjax/util.py:152-240)."""
import pytest
import torch

import blackjax_amd as bjx

pytestmark = pytest.mark.gpu

_FIELDS = ("logdensity", "acceptance_rate", "energy", "num_integration_steps", "num_trajectory_expansions",
           "is_divergent", "is_turning")


def _same_run(a, b):
    (sa, pa, ia), (sb, pb, ib) = a, b
    assert torch.equal(sa.position, sb.position)
    assert torch.equal(sa.logdensity, sb.logdensity)
    assert torch.equal(sa.logdensity_grad, sb.logdensity_grad)
    assert (pa is None) == (pb is None)
    if pa is not None:
        assert pa.shape == pb.shape and torch.equal(pa, pb)
    for f in _FIELDS:
        x, y = getattr(ia, f), getattr(ib, f)
        assert x.shape == y.shape, f
        if x.is_floating_point():
            assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), f
        else:
            assert torch.equal(x, y), f


@pytest.mark.parametrize("N,D,max_depth,eps", [
    (300, 64, 8, 0.1),      # busy phase -> speculative tail
    (9000, 16, 6, 0.2),     # busy phase -> the 8 192-row tier and every tier below it
    (17, 256, 7, 0.05),     # speculative tail from the first chunk on
])
def test_run_workspace_equals_per_call_driver(dev, monkeypatch, N, D, max_depth, eps):
    g = torch.Generator(device=dev)
    g.manual_seed(N + D)
    q0 = 0.2 * torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.NealFunnel()
    imm = torch.ones(D, device=dev)
    ws = bjx.nuts(fn, eps, imm, max_num_doublings=max_depth)
    ref = bjx.nuts(fn, eps, imm, max_num_doublings=max_depth)
    state = ws.init(q0)
    # (key, T, store_positions): the first call builds the workspace (capacity 512 without positions), the others
    # reuse it at other lengths; store_positions=True is its own workspace (capacity = T), rebuilt when T grows
    plan = [(1, 3, False), (2, 3, False), (3, 7, False), (4, 1, False), (5, 4, True), (6, 2, True), (7, 6, True),
            (8, 5, False)]
    for seed, T, keep in plan:
        key = bjx.random.key(seed)
        out_ws = ws.run(key, state, T, store_positions=keep)
        monkeypatch.setenv("BJX_NUTS_RUN_WS", "0")
        out_ref = ref.run(key, state, T, store_positions=keep)
        monkeypatch.delenv("BJX_NUTS_RUN_WS")
        _same_run(out_ws, out_ref)
        assert out_ws[2].logdensity.shape == (T, N)
        state = out_ws[0]


def test_run_workspace_equals_steps_and_survives_interleaving(dev):
    """run(T) on the workspace == T x step with run_inference_algorithm's keys; `step` calls (their own persistent
    workspace) interleaved with runs do not disturb either."""
    N, D, T = 200, 32, 5
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    fn = bjx.targets.NealFunnel()
    alg = bjx.nuts(fn, 0.1, torch.ones(D, device=dev), max_num_doublings=7)
    state0 = alg.init(0.3 * torch.randn(N, D, device=dev, generator=g))
    for rep in range(3):
        key = bjx.random.key(20 + rep)
        st_r, pos, info = alg.run(key, state0, T)
        st = state0
        for t, k in enumerate(bjx.random.split(key, T)):
            st, inf = alg.step(k, st)
            assert torch.equal(pos[t], st.position), (rep, t)
            assert torch.equal(info.num_integration_steps[t], inf.num_integration_steps)
            assert torch.equal(info.energy[t], inf.energy)
        assert torch.equal(st.position, st_r.position)
        state0 = st_r


def test_run_workspace_grows_past_its_capacity(dev):
    N, D = 64, 8
    fn = bjx.targets.NealFunnel()
    alg = bjx.nuts(fn, 0.2, torch.ones(D, device=dev), max_num_doublings=4)
    ref = bjx.nuts(fn, 0.2, torch.ones(D, device=dev), max_num_doublings=4, run_use_graph=False)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    state = alg.init(torch.randn(N, D, device=dev, generator=g))
    for T in (4, 600, 30):  # capacity 512 -> rebuilt at 600 -> reused
        key = bjx.random.key(T)
        _same_run(alg.run(key, state, T, store_positions=False), ref.run(key, state, T, store_positions=False))
