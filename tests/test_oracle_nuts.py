"""Pins the oracle's NUTS restatement against the reference's own tests (U-turn truth table,
expansion outcomes) and checks basic sampler statistics."""
import json
import os

import numpy as np
import pytest

from oracle import hmc as ohmc
from oracle import nuts as onuts
from oracle import prng, targets
from oracle.fp import f32

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_iterative_uturn_truth_table():
    k = KATS["iterative_uturn"]
    metric = ohmc.default_metric(np.ones(1, f32))
    r_ck = np.array(k["momentum_ckpts"], f32).reshape(-1, 1, 1)
    rs_ck = np.array(k["momentum_sum_ckpts"], f32).reshape(-1, 1, 1)
    r = np.full((1, 1), k["momentum"], f32)
    rs = np.full((1, 1), k["momentum_sum"], f32)
    for (idx_min, idx_max), expected in k["cases"]:
        assert onuts.is_iterative_turning(metric, r_ck, rs_ck, idx_min, idx_max, rs, r) == expected


def test_leaf_idx_to_ckpt_idxs():
    # termination.py:75-84 docstring examples: idx_max(6)=2, (7)=2, (13)=2 ; num_subtrees 6->0, 7->3, 13->1
    assert [onuts.leaf_idx_to_ckpt_idxs(n)[1] for n in (6, 7, 13)] == [2, 2, 2]
    assert [onuts.leaf_idx_to_ckpt_idxs(n)[1] - onuts.leaf_idx_to_ckpt_idxs(n)[0] + 1 for n in (6, 7, 13)] == [0, 3, 1]


def test_dynamic_expansion_outcomes():
    k = KATS["dynamic_expansion"]
    fn = targets.diag_gaussian(np.ones(1, f32))
    metric = ohmc.default_metric(np.ones(1, f32))
    key = prng.key(k["key_seed"])
    q = np.zeros((1, 1), f32)
    p = prng.normal(key, (1,))[None]
    lp, g = fn(q)
    z0 = ohmc.IntegratorState(q, p.astype(f32), lp, g)
    for eps, should_div, should_turn, doublings in k["cases"]:
        out = onuts._one_chain(key, z0, fn, f32(eps), metric, k["max_doublings"], k["divergence_threshold"])
        _, _, _, _, depth, n_states, _, div, turn = out
        assert (div, turn, depth) == (should_div, should_turn, doublings), (eps, div, turn, depth)
        if doublings == 10:
            assert n_states == 1023


@pytest.mark.parametrize("eps,should_diverge", [(0.0001, False), (1000.0, True)])
def test_progressive_integration_divergence(eps, should_diverge):
    """tests/mcmc/test_trajectory.py:20-75: standard normal, position 1.0, momentum = normal(key(0)), divergence
    threshold 1000 (the module constant there): a step of 1000 diverges at once, a step of 1e-4 never does (the
    reference integrates one subtree of at most 100 leaves; here the whole transition, depth limit 10)."""
    fn = targets.diag_gaussian(np.ones(1, f32))
    metric = ohmc.default_metric(np.ones(1, f32))
    key = prng.key(0)
    q = np.ones((1, 1), f32)
    p = prng.normal(key, (1,))[None].astype(f32)
    lp, g = fn(q)
    out = onuts._one_chain(key, ohmc.IntegratorState(q, p, lp, g), fn, f32(eps), metric, 10, 1000)
    _, _, _, _, depth, n_states, _, div, turn = out
    assert div is should_diverge or bool(div) == should_diverge
    if should_diverge:
        assert depth == 1 and n_states == 1  # the first leaf already diverges: nothing is added to the trajectory
    else:
        assert depth == 10 and n_states == 1023 and not turn


def test_nuts_statistics_normal():
    """reference tests/mcmc/test_sampling.py:1055-1187 flavour: N(1, 2^2), nuts eps=1."""

    def fn(q):
        g = -(q - f32(1.0)) / f32(4.0)
        return (0.5 * np.sum((q - f32(1.0)).astype(np.float64) * g, -1)).astype(f32), g.astype(f32)

    N = 48
    st = ohmc.init(np.ones((N, 1), f32), fn)
    draws, depths = [], []
    for t, kk in enumerate(prng.split(prng.key(12), 60)):
        st, info = onuts.kernel(kk, st, fn, f32(1.0), np.ones(1, f32))
        depths.append(info.num_trajectory_expansions)
        assert np.all(info.num_integration_steps >= 1)
        assert np.all((info.acceptance_rate >= 0) & (info.acceptance_rate <= 1 + 1e-6))
        if t >= 10:
            draws.append(st.position.copy())
    d = np.concatenate(draws)
    np.testing.assert_allclose(d.mean(), 1.0, atol=0.15)
    np.testing.assert_allclose(d.var(), 4.0, rtol=0.15)
    assert 1 <= np.mean(depths) <= 4
