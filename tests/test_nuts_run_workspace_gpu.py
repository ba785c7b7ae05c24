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


@pytest.mark.timeout(600)
def test_run_workspaces_interleaved_on_streams_and_threads(dev):
    """The purity contract (blackjax/base.py:24-85) against the run workspace: two `nuts` objects whose `run` and
    `step` calls alternate on two streams, then run concurrently from two threads, give the results of each alone."""
    import threading

    D = 32
    imm = torch.ones(D, device=dev)
    fn = bjx.targets.NealFunnel()
    gen = torch.Generator(device=dev).manual_seed(4)
    q0 = {"a": 0.3 * torch.randn(96, D, device=dev, generator=gen), "b": 0.3 * torch.randn(160, D, device=dev, generator=gen)}
    make = {"a": lambda: bjx.nuts(fn, 0.2, imm, max_num_doublings=6), "b": lambda: bjx.nuts(fn, 0.15, imm, max_num_doublings=5)}
    plan = [("run", 1, 4), ("step", 2, 0), ("run", 3, 2), ("run", 4, 6), ("step", 5, 0)]

    def drive(alg, st, items, stream=None):
        import contextlib

        out = []
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            for kind, seed, T in items:
                if kind == "run":
                    st, pos, info = alg.run(bjx.random.key(seed), st, T)
                    out.append((pos.clone(), info.num_integration_steps.clone(), info.energy.clone()))
                else:
                    st, info = alg.step(bjx.random.key(seed), st)
                    out.append((st.position.clone(), info.num_integration_steps.clone(), info.energy.clone()))
        return st, out

    def same(x, y):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            for s, t in zip(u, v):
                assert torch.equal(torch.nan_to_num(s.float(), nan=-7.0), torch.nan_to_num(t.float(), nan=-7.0))

    alone = {}
    for n in "ab":
        alg = make[n]()
        alone[n] = drive(alg, alg.init(q0[n]), plan)[1]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for s in (s1, s2):
        s.wait_stream(torch.cuda.current_stream(dev))
    stream_of = {"a": s1, "b": s2}
    algs = {n: make[n]() for n in "ab"}
    st, got = {}, {"a": [], "b": []}
    for n in "ab":
        with torch.cuda.stream(stream_of[n]):
            st[n] = algs[n].init(q0[n])
    for item in plan:
        for n in "ab":
            st[n], o = drive(algs[n], st[n], [item], stream_of[n])
            got[n] += o
    torch.cuda.synchronize()
    for n in "ab":
        same(got[n], alone[n])
    res, errs = {}, []

    def worker(n, stream):
        try:
            alg = make[n]()
            with torch.cuda.stream(stream):
                s0 = alg.init(q0[n])
            res[n] = drive(alg, s0, plan, stream)[1]
            stream.synchronize()
        except BaseException as e:
            errs.append(e)

    th = [threading.Thread(target=worker, args=("a", s1)), threading.Thread(target=worker, args=("b", s2))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a worker thread did not finish"
    assert not errs, errs
    torch.cuda.synchronize()
    for n in "ab":
        same(res[n], alone[n])


def test_run_workspace_per_chain_step_size_and_metric(dev, monkeypatch):
    """Per-chain step sizes and a per-chain diagonal metric (what window_adaptation hands to the sampler) go through the
    workspace's static copies; a multi-stage integrator runs on it too."""
    N, D = 150, 24
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    fn = bjx.targets.NealFunnel()
    eps = 0.05 + 0.1 * torch.rand(N, device=dev, generator=g)
    imm = 0.5 + torch.rand(N, D, device=dev, generator=g)
    q0 = 0.3 * torch.randn(N, D, device=dev, generator=g)
    for integrator in (bjx.integrators.velocity_verlet, bjx.integrators.mclachlan):
        ws = bjx.nuts(fn, eps, imm, max_num_doublings=6, integrator=integrator)
        ref = bjx.nuts(fn, eps, imm, max_num_doublings=6, integrator=integrator)
        state = ws.init(q0)
        for seed, T in ((1, 3), (2, 5), (3, 2)):
            key = bjx.random.key(seed)
            out_ws = ws.run(key, state, T)
            monkeypatch.setenv("BJX_NUTS_RUN_WS", "0")
            out_ref = ref.run(key, state, T)
            monkeypatch.delenv("BJX_NUTS_RUN_WS")
            _same_run(out_ws, out_ref)
            state = out_ws[0]
