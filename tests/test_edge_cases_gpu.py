"""Edge cases at the boundary: empty batches, zero-length trajectories, one-dimensional targets,
automatic chain blocking."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd.hmc import auto_chain_block
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
f32 = np.float32


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def test_empty_batch_everywhere(dev):
    D = 12
    inv_var = torch.ones(D, device=dev)
    fn = bjx.targets.DiagGaussian(inv_var)
    q0 = torch.empty(0, D, device=dev)
    for L in (0, 3):
        alg = bjx.hmc(fn, 0.1, torch.ones(D, device=dev), L)
        st = alg.init(q0)
        st2, info = alg.step(prng.key(0), st)
        assert st2.position.shape == (0, D) and info.acceptance_rate.shape == (0,)
        assert info.proposal.position.shape == (0, D)
    dh = bjx.dynamic_hmc(fn, 0.1, torch.ones(D, device=dev))
    st = dh.init(q0, prng.key(1))
    assert st.random_generator_arg.shape == (0, 2)
    nuts = bjx.nuts(fn, 0.1, torch.ones(D, device=dev), max_num_doublings=3)
    st = nuts.init(q0)
    st2, info = nuts.step(prng.key(2), st)
    assert st2.position.shape == (0, D) and info.num_integration_steps.shape == (0,)
    st3, positions, rinfo = nuts.run(prng.key(3), st, 4)
    assert positions.shape == (4, 0, D) and rinfo.acceptance_rate.shape == (4, 0)
    mh = bjx.mhmc(fn, 0.1, torch.ones(D, device=dev), 3)
    st = mh.init(q0)
    st2, info = mh.step(prng.key(4), st)
    assert st2.position.shape == (0, D) and info.acceptance_rate.shape == (0,)


@pytest.mark.parametrize("N,D", [(9, 1), (5, 3)])
def test_zero_length_trajectory_and_tiny_dims(dev, N, D):
    """L = 0: the proposal is the initial state, delta = 0, always accepted (trajectory.py:155-165
    with an empty loop); D = 1 exercises the scalar (non-float4) path end to end."""
    inv_var = np.linspace(0.5, 2.0, D).astype(f32)
    fn_o, fn_g = otargets.diag_gaussian(inv_var), bjx.targets.DiagGaussian(dev_t(inv_var, dev))
    q0 = prng.normal(prng.key(2), (N, D))
    imm = np.linspace(1.0, 3.0, D).astype(f32)
    for L in (0, 1, 4):
        st_o = ohmc.init(q0, fn_o)
        alg = bjx.hmc(fn_g, 0.3, dev_t(imm, dev), L)
        st_g = alg.init(dev_t(q0, dev))
        for k in prng.split(prng.key(5), 3):
            st_o, info_o = ohmc.kernel(k, st_o, fn_o, f32(0.3), imm, L)
            st_g, info_g = alg.step(k, st_g)
            assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
            assert np.array_equal(t2n(st_g.position), st_o.position)
            np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
            assert np.array_equal(t2n(info_g.momentum), info_o.momentum)
        if L == 0:
            assert bool(info_g.is_accepted.all()) and float(info_g.acceptance_rate.min()) == 1.0


def test_auto_chain_block(dev):
    assert auto_chain_block(65536, 1024) == 16384
    assert auto_chain_block(32768, 4096) == 4096
    assert auto_chain_block(65536, 256) == 65536  # the whole batch fits: one block
    assert auto_chain_block(1000, 1024) == 1000
    assert auto_chain_block(100000, 3) == 100000
    # blocked == unblocked bit for bit, ragged last block (40 000 = 2 x 16 384 + 7 232)
    N, D, L = 40000, 1024, 3
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    q0 = torch.randn(N, D, device=dev, generator=g)
    fn = bjx.targets.DiagGaussian(torch.ones(D, device=dev))
    imm = torch.ones(D, device=dev)
    a = bjx.hmc(fn, 0.05, imm, L)
    b = bjx.hmc(fn, 0.05, imm, L, chain_block="auto")
    sa, sb = a.init(q0), b.init(q0)
    for k in prng.split(prng.key(8), 2):
        sa, ia = a.step(k, sa)
        sb, ib = b.step(k, sb)
    assert torch.equal(sa.position, sb.position) and torch.equal(ia.is_accepted, ib.is_accepted)
    assert torch.equal(ia.proposal.position, ib.proposal.position)
    assert torch.equal(ia.energy, ib.energy)


def test_pytree_positions_through_ravel(dev):
    """A log-density written against a dict of parameters (the reference's usual input) runs through
    ravel_chain_pytree / flat_logdensity and gives the draws of the equivalent flat callable."""
    N = 64
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    tree = {"loc": torch.randn(N, 5, device=dev, generator=g), "log_scale": 0.1 * torch.randn(N, device=dev, generator=g)}

    def logdensity(p):
        return -0.5 * (p["loc"] ** 2).sum(-1) * torch.exp(-2 * p["log_scale"]) - 0.5 * p["log_scale"] ** 2

    def logdensity_flat(q):  # same function, written against the flat layout [loc(5), log_scale(1)]
        return -0.5 * (q[:, :5] ** 2).sum(-1) * torch.exp(-2 * q[:, 5]) - 0.5 * q[:, 5] ** 2

    flat, unravel = bjx.util.ravel_chain_pytree(tree)
    a = bjx.nuts(bjx.util.flat_logdensity(logdensity, unravel), 0.2, torch.ones(6, device=dev), max_num_doublings=4)
    b = bjx.nuts(logdensity_flat, 0.2, torch.ones(6, device=dev), max_num_doublings=4)
    sa, sb = a.init(flat), b.init(flat)
    for k in prng.split(prng.key(8), 3):
        sa, ia = a.step(k, sa)
        sb, ib = b.step(k, sb)
        assert torch.equal(ia.num_integration_steps, ib.num_integration_steps)
        assert torch.allclose(sa.position, sb.position, atol=1e-6)
    assert unravel(sa.position)["loc"].shape == (N, 5)


# ---------------------------------------------------------------------------------------------- re-entrancy
def _steps(alg, st, keys, stream=None):
    """`len(keys)` steps; returns the states / infos as CPU tensors (position, logdensity, the per-chain info scalars)."""
    import contextlib

    out = []
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        for k in keys:
            st, info = alg.step(k, st)
            out.append((st.position.clone(), st.logdensity.clone(), info.acceptance_rate.clone(),
                        info.num_integration_steps.clone() if torch.is_tensor(info.num_integration_steps) else None))
    return st, out


def _same_runs(a, b):
    assert len(a) == len(b)
    for (pa, la, aa, na), (pb, lb, ab, nb) in zip(a, b):
        assert torch.equal(pa, pb) and torch.equal(la, lb) and torch.equal(aa, ab)
        assert (na is None and nb is None) or torch.equal(na, nb)


@pytest.mark.timeout(600)
def test_algorithm_objects_interleaved_on_streams_and_threads(dev, monkeypatch):
    """The purity contract of base.py:24-85 -- `step(rng_key, state)` is a function of its arguments -- against the
    state the drivers keep behind it (persistent NUTS workspaces keyed per algorithm object, shape and stream; recorded
    graphs; caches; BJX_* switches read at call time): two `nuts` objects and one `hmc` object stepped ALTERNATELY on two
    streams, then concurrently from two Python threads, then with the environment switches changed in mid-run, give
    bit for bit the results of each object stepped alone."""
    import threading

    D = 32
    imm = torch.ones(D, device=dev)
    funnel = bjx.targets.NealFunnel()
    gauss = bjx.targets.DiagGaussian(torch.linspace(0.5, 2.0, D, device=dev))
    gen = torch.Generator(device=dev).manual_seed(11)
    q_a = 0.3 * torch.randn(96, D, device=dev, generator=gen)
    q_b = 0.3 * torch.randn(160, D, device=dev, generator=gen)
    q_c = torch.randn(200, D, device=dev, generator=gen)
    make = {"a": lambda: bjx.nuts(funnel, 0.2, imm, max_num_doublings=6),
            "b": lambda: bjx.nuts(funnel, 0.15, imm, max_num_doublings=5),
            "c": lambda: bjx.hmc(gauss, 0.2, imm, 5)}
    q0 = {"a": q_a, "b": q_b, "c": q_c}
    keys = list(bjx.random.split(bjx.random.key(3), 5))
    # each object alone, on the default stream
    alone = {}
    for n in "abc":
        alg = make[n]()
        alone[n] = _steps(alg, alg.init(q0[n]), keys)[1]
    torch.cuda.synchronize()

    # (1) alternately, on two streams (a and c share one)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s1.wait_stream(torch.cuda.current_stream(dev))
    s2.wait_stream(torch.cuda.current_stream(dev))
    algs = {n: make[n]() for n in "abc"}
    stream_of = {"a": s1, "b": s2, "c": s1}
    st, got = {}, {n: [] for n in "abc"}
    for n in "abc":
        with torch.cuda.stream(stream_of[n]):
            st[n] = algs[n].init(q0[n])
    for k in keys:
        for n in "abc":
            st[n], o = _steps(algs[n], st[n], [k], stream_of[n])
            got[n] += o
    torch.cuda.synchronize()
    for n in "abc":
        _same_runs(got[n], alone[n])

    # (2) concurrently from two Python threads, each with its own stream and its own objects
    res, errs = {}, []

    def worker(names, stream):
        try:
            for n in names:
                alg = make[n]()
                with torch.cuda.stream(stream):
                    s0 = alg.init(q0[n])
                res[n] = _steps(alg, s0, keys, stream)[1]
            stream.synchronize()
        except BaseException as e:  # surfaced in the main thread
            errs.append(e)

    th = [threading.Thread(target=worker, args=(("a", "c"), s1)), threading.Thread(target=worker, args=(("b",), s2))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a worker thread did not finish"
    assert not errs, errs
    torch.cuda.synchronize()
    for n in "abc":
        _same_runs(res[n], alone[n])

    # (3) the BJX_* switches the drivers read at call time, changed in mid-run: scheduling only, never results
    algs = {n: make[n]() for n in "abc"}
    st = {n: algs[n].init(q0[n]) for n in "abc"}
    got = {n: [] for n in "abc"}
    settings = [{}, {"BJX_NUTS_SPEC_ROWS": "16", "BJX_CHAIN_BLOCK": "64"}, {"BJX_NUTS_SYNC_EVERY": "2", "BJX_NUTS_TAIL_ROWS": "64"},
                {"BJX_NUTS_STEP_DRIVER": "lockstep"}, {}]
    names = ("BJX_NUTS_SPEC_ROWS", "BJX_CHAIN_BLOCK", "BJX_NUTS_SYNC_EVERY", "BJX_NUTS_TAIL_ROWS", "BJX_NUTS_STEP_DRIVER")
    for k, env in zip(keys, settings):
        for name in names:
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        for n in "abc":
            st[n], o = _steps(algs[n], st[n], [k])
            got[n] += o
    torch.cuda.synchronize()
    for n in "abc":
        _same_runs(got[n], alone[n])
