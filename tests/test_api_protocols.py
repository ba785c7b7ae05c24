"""The reference's protocol checks (tests/test_api_protocols.py:176-232) for the samplers this engine
exposes: the factory returns a SamplingAlgorithm, ``init``'s first parameter is ``position``, ``step``'s
first two are ``rng_key, state``; the GPU half does the init -> step round trip with the reference's
calling conventions (``init(position)`` or, for dhmc / dmhmc / ghmc, ``init(position, rng_key)``)."""
import inspect

import pytest
import torch

import blackjax_amd as bjx

_NEEDS_RNG_KEY = {"ghmc", "dhmc", "dmhmc"}
_ALGORITHMS = ["hmc", "nuts", "mhmc", "dhmc", "dmhmc", "ghmc"]


def _make(name, fn, inv_mass):
    if name in ("hmc", "mhmc"):
        return getattr(bjx, name)(fn, step_size=0.1, inverse_mass_matrix=inv_mass, num_integration_steps=10)
    if name == "nuts":
        return bjx.nuts(fn, step_size=0.1, inverse_mass_matrix=inv_mass)
    if name in ("dhmc", "dmhmc"):
        return getattr(bjx, name)(fn, step_size=0.1, inverse_mass_matrix=inv_mass)
    return bjx.ghmc(fn, step_size=0.1, momentum_inverse_scale=inv_mass, alpha=0.5, delta=0.5)


def _std_normal(q):
    return -0.5 * (q * q).sum(-1)


@pytest.mark.parametrize("name", _ALGORITHMS)
def test_factory_and_signatures(name):
    alg = _make(name, _std_normal, torch.ones(3))
    assert isinstance(alg, bjx.SamplingAlgorithm)
    init, step = alg  # unpacks like the reference's two-field NamedTuple
    assert init is alg.init and step is alg.step
    assert list(inspect.signature(alg.init).parameters)[0] == "position"
    assert list(inspect.signature(alg.step).parameters)[:2] == ["rng_key", "state"]


def test_aliases_and_families():
    assert bjx.multinomial_hmc is bjx.mhmc and bjx.dhmc is bjx.dynamic_hmc  # blackjax/__init__.py:152, 117
    assert bjx.hmc_family == [bjx.hmc, bjx.nuts, bjx.mhmc]
    for api in (bjx.hmc, bjx.nuts, bjx.mhmc, bjx.dhmc, bjx.dmhmc, bjx.ghmc):
        assert callable(api.init) and callable(api.build_kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _ALGORITHMS)
def test_init_step_roundtrip(dev, name):
    alg = _make(name, _std_normal, torch.ones(3, device=dev))
    init_key, step_key = bjx.random.split(bjx.random.key(0), 2)
    position = torch.full((5, 3), 0.5, device=dev)
    state = alg.init(position, init_key) if name in _NEEDS_RNG_KEY else alg.init(position)
    new_state, info = alg.step(step_key, state)
    assert new_state.position.shape == (5, 3) and bool(torch.isfinite(new_state.logdensity).all())
    assert info.acceptance_rate.shape == (5,)


@pytest.mark.gpu
def test_logdensity_evaluations_per_transition(dev):
    """The analogue of the reference's tests/test_compilation.py (how often the log-density is traced):
    how often the user's callable is EVALUATED -- once at init, then exactly once per leapfrog of a
    transition (per chain block), with no hidden extra calls (an undeclared callable is never recorded
    into a HIP graph, so there are no warm-up evaluations either)."""
    calls = {"n": 0}

    @bjx.no_trace  # eager autograd on every call (the default for a plain function is tested below)
    def fn(q):
        calls["n"] += 1
        return -0.5 * (q * q).sum(-1)

    q0 = torch.randn(64, 6, device=dev)
    ones = torch.ones(6, device=dev)
    for make, per_step in (
        (lambda: bjx.hmc(fn, 0.1, ones, 7), 7),
        (lambda: bjx.hmc(fn, 0.1, ones, 7, chain_block=16), 7 * 4),
        (lambda: bjx.mhmc(fn, 0.1, ones, 5), 5),
        (lambda: bjx.ghmc(fn, 0.1, ones, 0.5, 0.2), 1),
    ):
        alg = make()
        calls["n"] = 0
        try:
            state = alg.init(q0)
        except ValueError:  # ghmc draws its momentum and slice at init: it needs a key (raised before any evaluation)
            state = alg.init(q0, bjx.random.key(1))
        assert calls["n"] == 1
        for k in bjx.random.split(bjx.random.key(2), 3):
            state, _ = alg.step(k, state)
        assert calls["n"] == 1 + 3 * per_step, (per_step, calls["n"])


@pytest.mark.gpu
def test_a_plain_function_is_evaluated_once_and_traced_once(dev):
    """The reference's tests/test_compilation.py:19-100 asks that the log-density be TRACED at most twice per kernel:
    here a plain PyTorch function is evaluated once (init, under autograd), traced once (torch.fx -> generated
    value-and-gradient kernel, checked against that evaluation) and its Python code does not run again -- except for the
    sparse re-checks of the generated kernel against the live function (calls 16, 256, every 4 096th: one evaluation each)."""
    calls = {"n": 0}

    def fn(q):
        calls["n"] += 1
        return -0.5 * (q * q).sum(-1)

    q0 = torch.randn(64, 8, device=dev)
    alg = bjx.hmc(fn, 0.1, torch.ones(8, device=dev), 7)
    state = alg.init(q0)
    assert calls["n"] == 2
    state, info = alg.step(bjx.random.key(2), state)  # 7 gradient evaluations: kernel calls 1 .. 7
    assert calls["n"] == 2 and bool(info.is_accepted.any())
    for k in bjx.random.split(bjx.random.key(3), 3):    # kernel calls 8 .. 28: the re-check at call 16 evaluates fn once
        state, info = alg.step(k, state)
    assert calls["n"] == 3


@pytest.mark.gpu
def test_a_traced_function_that_goes_stale_is_caught_and_put_back_on_autograd(dev):
    """A traced function is the function as it was at its first call (as under jax.jit).  One that closes over Python
    state which changes later is caught by the re-check (call 16 here) and evaluated eagerly from then on, with a
    RuntimeWarning -- never silently stale for more than a re-check interval."""
    from blackjax_amd import _util

    box = {"beta": 1.0}

    def tempered(q):
        return -0.5 * box["beta"] * (q * q).sum(-1)

    q = torch.randn(32, 16, device=dev)
    vg = _util.value_and_grad(tempered)
    lp, g = vg(q)                                # first call: autograd, then traced with beta = 1
    assert torch.allclose(g, -q) and list(vg._bjx_elementwise.values())[0] is not None
    box["beta"] = 3.0                            # the user changes the temperature: the kernel still has beta = 1 ...
    for _ in range(15):                          # kernel calls 1 .. 15
        lp, g = vg(q)
        assert torch.allclose(g, -q)
    with pytest.warns(RuntimeWarning, match="no longer agrees"):
        lp, g = vg(q)                            # ... until kernel call 16 re-checks it against the live function
    assert torch.allclose(g, -3.0 * q) and list(vg._bjx_elementwise.values())[0] is None
    lp, g = vg(q)
    assert torch.allclose(g, -3.0 * q)           # eager autograd from here on
