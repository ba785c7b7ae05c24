"""The oracle (CPU) and the HIP path (GPU) against THE REFERENCE'S OWN SOURCE, executed on a stand-in for JAX.

``tests/golden/ref_shim_fixtures.json`` is written by ``tests/golden/gen_ref_shim_fixtures.py``: ``/root/reference/blackjax``
imported unmodified on top of ``tests/refshim`` (torch / NumPy stand-in for the ~60 JAX functions the hot path calls --
NOT JAX, read its docstring) and run: ``blackjax.hmc / mhmc / dynamic_hmc / nuts / ghmc`` transitions (diagonal and dense
metrics, all four integrators, rejections, divergences, depth limits), ``run_inference_algorithm``, ``build_schedule``,
four ``window_adaptation`` runs with every step's adaptation state, a ``chees_adaptation`` and a ``meads_adaptation`` run
and the diagnostics (``effective_sample_size``, ``rhat``, ``ess_bulk``, ``ess_tail``).  What this pins: everything the reference's code DECIDES
(key consumption, tree growth and termination, acceptance, adaptation updates, window ends) and its arithmetic up to fp32
rounding.  What it does not: the ``jax.random`` bit streams (the stand-in's ``jax.random`` is ``oracle/prng.py``: SURVEY row
a34 stays "parity unpinned").

Discrete results must be EQUAL (accept bits, divergence flags, leapfrog counts, tree depths, turning flags, dual-averaging
step counters, Welford counts); floating-point results agree to the tolerance written at each assert (two fp32
implementations of the same expressions).  The warm-ups are compared STEP BY STEP from the reference's own state: an
adaptive run amplifies a one-ulp difference by orders of magnitude within ~20 steps, whoever computes it.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import adaptation as oad
from oracle import ghmc as oghmc
from oracle import hmc as ohmc
from oracle import integrators as oint
from oracle import nuts as onuts
from oracle import prng
from oracle import targets as otargets

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "ref_shim_fixtures.json")
f32 = np.float32

with open(PATH) as _fh:
    FX = json.load(_fh)  # the stand-in's set: committed, defines the case names
# the same cases from a REAL JAX + BlackJAX (BJX_REAL_JAX=1 python tests/golden/gen_ref_shim_fixtures.py on a machine that
# has them): absent so far.  When present every comparison below runs against it as well -- and then jax.random IS checked.
JAX_PATH = os.path.join(HERE, "golden", "ref_jax_fixtures.json")
_SETS = {"reference-on-stand-in": FX}
if os.path.exists(JAX_PATH):
    with open(JAX_PATH) as _fh:
        _SETS["reference-on-real-jax"] = json.load(_fh)


@pytest.fixture(params=sorted(_SETS))
def fx(request):
    return _SETS[request.param]


def unhex(x):
    return np.asarray(x, dtype=np.uint32).view(f32)


def ladder(D, lo, hi):
    return (10.0 ** (lo + (hi - lo) * np.arange(D) / max(D - 1, 1))).astype(f32)


def oracle_target(t, D):
    if t["kind"] == "diag_gaussian":
        s = ladder(D, t["lo"], t["hi"])
        return otargets.diag_gaussian((f32(1) / (s * s)).astype(f32))
    if t["kind"] == "funnel":
        return otargets.neal_funnel()
    return otargets.ar1_gaussian(t["rho"], D)


def initial_positions(c, N, D):
    q = prng.normal(prng.key(c["q0_key_seed"]), (N, D))
    scale = c.get("q0_scale")
    if scale == "sigma":
        return (ladder(D, c["target"]["lo"], c["target"]["hi"]) * q).astype(f32)
    return q if scale is None else (f32(scale) * q).astype(f32)


def metric_of(c, D):
    if c["metric"] == "identity":
        return np.ones(D, f32)
    if c["metric"] == "ladder":
        s = ladder(D, c["target"]["lo"], c["target"]["hi"])
        return (s * s).astype(f32)
    return otargets.ar1_covariance(c["metric_rho"], D)


def test_fixture_provenance():
    assert FX["generator"] == "tests/golden/gen_ref_shim_fixtures.py" and "NOT produced by JAX" in FX["what"]
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("/root/reference is not on this box: the source hashes cannot be re-checked")
    import hashlib

    for rel, digest in FX["reference_sources_sha256"].items():
        with open(os.path.join(ref, rel), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == digest, f"{rel} changed since the fixtures were generated"


def _check_sampler(c, new_position, info, nuts, dynamic_arg=None):
    """``new_position`` / ``info`` from the implementation under test (NumPy arrays), ``c`` the reference's record."""
    rows = c["rows"]
    assert np.array_equal(np.asarray(info["is_divergent"]).astype(int), c["is_divergent"])
    assert np.array_equal(np.broadcast_to(np.asarray(info["num_integration_steps"]), (c["N"],)), c["num_integration_steps"])
    if nuts:
        assert np.array_equal(info["num_trajectory_expansions"], c["num_trajectory_expansions"])
        assert np.array_equal(np.asarray(info["is_turning"]).astype(int), c["is_turning"])
    else:
        assert np.array_equal(np.asarray(info["is_accepted"]).astype(int), c["is_accepted"])
    # floating point: two fp32 implementations of the same expressions (the reference's dense products are plain fp32
    # dots, the oracle's are fp64-accumulated: 1e-5 absolute there, ~1e-6 otherwise)
    np.testing.assert_allclose(info["acceptance_rate"], unhex(c["acceptance_rate"]), rtol=0, atol=5e-4)
    ref_e = unhex(c["energy"])
    fin = np.isfinite(ref_e)
    assert np.array_equal(np.isfinite(info["energy"]), fin)
    np.testing.assert_allclose(np.asarray(info["energy"])[fin], ref_e[fin], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(np.asarray(info["momentum"])[rows], unhex(c["momentum"]), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(np.asarray(new_position)[rows], unhex(c["position"]), rtol=2e-4, atol=3e-5)
    if nuts:
        np.testing.assert_allclose(np.asarray(info["leftmost_position"])[rows], unhex(c["leftmost_position"]), rtol=2e-4, atol=3e-5)
        np.testing.assert_allclose(np.asarray(info["rightmost_position"])[rows], unhex(c["rightmost_position"]), rtol=2e-4, atol=3e-5)
    else:
        ref_p = unhex(c["proposal_position"])
        ok = np.isfinite(ref_p).all(-1)
        np.testing.assert_allclose(np.asarray(info["proposal_position"])[rows][ok], ref_p[ok], rtol=2e-4, atol=3e-5)
    if dynamic_arg is not None:
        assert np.array_equal(dynamic_arg, np.asarray(c["next_random_generator_arg"], np.uint32))


def _coefficients(c):
    name = c.get("integrator", "velocity_verlet")
    return None if name == "velocity_verlet" else getattr(oint, name)


@pytest.mark.parametrize("name", sorted(FX["samplers"]))
def test_oracle_transition_equals_the_reference_code(fx, name):
    c = fx["samplers"][name]
    N, D = c["N"], c["D"]
    fn, q0, imm = oracle_target(c["target"], D), initial_positions(c, N, D), metric_of(c, D)
    key = np.asarray(c["step_key"], np.uint32)
    thr, coef, algo = c.get("divergence_threshold", 1000), _coefficients(c), c["algorithm"]
    dyn = None
    with np.errstate(over="ignore", invalid="ignore"):
        if algo == "hmc":
            st, info = ohmc.kernel(key, ohmc.init(q0, fn), fn, f32(c["eps"]), imm, c["L"], thr, coefficients=coef)
        elif algo == "mhmc":
            st, info = ohmc.mhmc_kernel(key, ohmc.init(q0, fn), fn, f32(c["eps"]), imm, c["L"], thr, coefficients=coef)
        elif algo == "nuts":
            st, info = onuts.kernel(key, ohmc.init(q0, fn), fn, f32(c["eps"]), imm, c["max_num_doublings"], thr,
                                    coefficients=coef)
        else:  # dynamic_hmc / dmhmc (dynamic trajectory lengths; dmhmc: with the multinomial proposal)
            s0 = ohmc.init(q0, fn)
            ds = ohmc.DynamicHMCState(s0.position, s0.logdensity, s0.logdensity_grad,
                                      prng.split(prng.key(c["arg_key_seed"]), N))
            st, info = ohmc.dynamic_hmc_kernel(key, ds, fn, f32(c["eps"]), imm, thr, multinomial=algo == "dmhmc")
            dyn = st.random_generator_arg
    d = info._asdict()
    if algo == "nuts":
        d["leftmost_position"] = info.trajectory_leftmost_state.position
        d["rightmost_position"] = info.trajectory_rightmost_state.position
    else:
        d["proposal_position"] = info.proposal.position
    _check_sampler(c, st.position, d, algo == "nuts", dyn)


def test_cases_cover_what_they_are_named_for():
    s = FX["samplers"]
    assert 0 < sum(s["hmc_rejections"]["is_accepted"]) < s["hmc_rejections"]["N"]
    assert 0 < sum(s["hmc_divergent"]["is_divergent"]) < s["hmc_divergent"]["N"]
    assert sum(s["hmc_all_divergent"]["is_divergent"]) == s["hmc_all_divergent"]["N"]
    assert 0 < sum(s["nuts_divergent"]["is_divergent"]) < s["nuts_divergent"]["N"]
    assert any(n not in (1, 3, 7, 15, 31, 63) for n in s["nuts_divergent"]["num_integration_steps"])  # stopped mid-subtree
    assert max(s["nuts_funnel_deep"]["num_trajectory_expansions"]) == s["nuts_funnel_deep"]["max_num_doublings"]
    assert set(s["nuts_depth_limit_2"]["num_integration_steps"]) == {3} and not any(s["nuts_depth_limit_2"]["is_turning"])
    assert set(s["nuts_depth_limit_10"]["num_integration_steps"]) == {1023} and not any(s["nuts_depth_limit_10"]["is_turning"])
    assert len(set(s["dynamic_hmc"]["num_integration_steps"])) > 3
    assert len(set(s["nuts_funnel"]["num_integration_steps"])) > 1


def test_schedules_equal_the_reference_code(fx):
    """staged_adaptation.py:366-403 executed, 16 lengths -- the oracle's AND the product's ``build_schedule``."""
    from blackjax_amd.adaptation import build_schedule as product_schedule

    for T, ref in fx["schedules"].items():
        assert [[int(a), int(bool(b))] for a, b in oad.build_schedule(int(T))] == ref, T
        assert [[int(a), int(bool(b))] for a, b in product_schedule(int(T))] == ref, T


def test_run_inference_key_layout_equals_the_reference_code(fx):
    """Step-major keys (scan over ``split(key, T)`` of vmap over ``split(keys[t], N)``), and the reference's own
    ``run_inference_algorithm(initial_position=...)`` on one chain (util.py:198-203: the key is split once more first)."""
    r = fx["run_inference"]
    N, D, L, T = r["N"], r["D"], r["L"], r["T"]
    fn = oracle_target(r["target"], D)
    q0 = prng.normal(prng.key(r["q0_key_seed"]), (N, D))
    _, pos, infos = ohmc.run(prng.key(r["run_key_seed"]), ohmc.init(q0, fn), fn, f32(r["eps"]), np.ones(D, f32), L, T)
    assert np.array_equal(np.stack([i.is_accepted for i in infos]).astype(int), r["is_accepted"])
    np.testing.assert_allclose(pos, unhex(r["positions"]), rtol=1e-5, atol=5e-6)
    run_key = prng.split(prng.key(r["single_chain_key_seed"]), 2)[0]  # rng_key, init_key = split(rng_key, 2)
    keys = prng.split(run_key, T)
    st, P, A = ohmc.init(q0[:1], fn), [], []
    for t in range(T):
        st, inf = ohmc.kernel(None, st, fn, f32(r["eps"]), np.ones(D, f32), L, chain_keys_override=keys[t:t + 1])
        P.append(st.position[0])
        A.append(int(inf.is_accepted[0]))
    assert A == r["single_chain_is_accepted"]
    np.testing.assert_allclose(np.stack(P), unhex(r["single_chain_positions"]), rtol=1e-5, atol=5e-6)


def _initial_imm(c, D):
    if c.get("initial_imm") == "ladder":
        s0 = ladder(D, -0.3, 0.3)
        return (s0 * s0).astype(f32)
    return None


def _warmup_arrays(c):
    names = ("log_step_size", "log_step_size_avg", "avg_error", "mu", "step_size", "inverse_mass_matrix", "welford_mean", "welford_m2")
    return {k: unhex(c[k]) for k in names}, np.asarray(c["da_step"]), np.asarray(c["welford_n"])


@pytest.mark.parametrize("name", sorted(FX["warmup"]))
def test_adaptation_updates_equal_the_reference_code_step_by_step(fx, name):
    """Every step of the reference's run: its state at t, its new position and acceptance rate -> the oracle's
    ``adapt_update`` -> its state at t + 1 (dual averaging, Welford, window ends with shrinkage, re-initialisation)."""
    c = fx["warmup"][name]
    N, D, T, diag = c["N"], c["D"], c["T"], c["diag"]
    pos, acc = unhex(c["position"]), unhex(c["acceptance_rate"])
    L, step, wn = _warmup_arrays(c)
    ws = oad.adapt_init(N, D, c.get("initial_step_size", 1.0), is_diag=diag, initial_imm=_initial_imm(c, D))
    n_ends = 0
    for t, (stage, end) in enumerate(oad.build_schedule(T)):
        n_ends += int(bool(end))
        new = oad.adapt_update(ws, stage, end, pos[:, t], acc[:, t], target=c.get("target_acceptance_rate", 0.8),
                               is_diag=diag, shrinkage=c.get("shrinkage", 0.0))
        assert np.all(np.asarray(new.ss_state.step) == step[:, t]), t
        assert np.all(np.asarray(new.imm_state.wc_state.sample_size) == wn[:, t]), t
        got = {"log_step_size": new.ss_state.log_step_size, "log_step_size_avg": new.ss_state.log_step_size_avg,
               "avg_error": new.ss_state.avg_error, "mu": new.ss_state.mu, "step_size": new.step_size,
               "inverse_mass_matrix": new.inverse_mass_matrix, "welford_mean": new.imm_state.wc_state.mean,
               "welford_m2": new.imm_state.wc_state.m2}
        for k, v in got.items():  # measured: 0 for six of the eight, one ulp for step_size / m2 (5e-6: dense m2)
            np.testing.assert_allclose(np.asarray(v), L[k][:, t], rtol=5e-6, atol=1e-7, err_msg=f"{k} at step {t}")
        ws = oad.StagedAdaptationState(  # continue from the REFERENCE's state
            oad.DualAveragingState(L["log_step_size"][:, t], L["log_step_size_avg"][:, t], int(step[0, t]),
                                   L["avg_error"][:, t], L["mu"][:, t]),
            oad.MassMatrixState(L["inverse_mass_matrix"][:, t],
                                oad.WelfordState(L["welford_mean"][:, t], L["welford_m2"][:, t], int(wn[0, t]))),
            L["step_size"][:, t], L["inverse_mass_matrix"][:, t])
    assert n_ends >= 1
    # final(): step size = exp(log_step_size_avg), the metric as it stands (staged_adaptation.py:301-305)
    fin = np.exp(L["log_step_size_avg"][:, -1].astype(np.float64)).astype(f32)
    np.testing.assert_allclose(unhex(c["final_step_size"]), fin, rtol=2e-7)
    assert np.array_equal(unhex(c["final_inverse_mass_matrix"]), L["inverse_mass_matrix"][:, -1])


@pytest.mark.parametrize("name", sorted(FX["warmup"]))
def test_warmup_transitions_equal_the_reference_code_step_by_step(fx, name):
    """The transition inside each warm-up step: chain key ``split(run_key, N)[c]``, step key ``split(chain_key, T)[t]``
    (staged_adaptation.py:868), the step size and metric of the state BEFORE the step -- from the reference's position."""
    c = fx["warmup"][name]
    N, D, T, diag = c["N"], c["D"], c["T"], c["diag"]
    fn = oracle_target(c["target"], D)
    q = initial_positions(c, N, D)
    pos, acc = unhex(c["position"]), unhex(c["acceptance_rate"])
    eps, imm = unhex(c["step_size"]), unhex(c["inverse_mass_matrix"])
    chain_keys = prng.split(np.asarray(c["run_key"], np.uint32), N)
    moved = 0
    for ci in range(N):
        keys = prng.split(chain_keys[ci], T)
        for t in range(0, T, 1 if T <= 100 else 2):
            q_t = q[ci:ci + 1] if t == 0 else pos[ci:ci + 1, t - 1]
            e_t = f32(c.get("initial_step_size", 1.0)) if t == 0 else eps[ci, t - 1]
            m0 = _initial_imm(c, D)
            m_t = (m0 if m0 is not None else (np.ones(D, f32) if diag else np.eye(D, dtype=f32))) if t == 0 else imm[ci, t - 1]
            st = ohmc.init(q_t, fn)
            with np.errstate(over="ignore", invalid="ignore"):
                if c["algorithm"] == "hmc":
                    st2, inf = ohmc.kernel(None, st, fn, e_t, m_t, c["L"], chain_keys_override=keys[t:t + 1])
                else:
                    st2, inf = onuts.kernel(None, st, fn, e_t, m_t, c["max_num_doublings"], chain_keys_override=keys[t:t + 1])
            np.testing.assert_allclose(st2.position[0], pos[ci, t], rtol=1e-4, atol=2e-5, err_msg=f"chain {ci} step {t}")
            assert abs(float(inf.acceptance_rate[0]) - float(acc[ci, t])) < 1e-4, (ci, t)
            moved += int(not np.array_equal(pos[ci, t], q_t[0]))
    assert moved > 10


def test_ghmc_equals_the_reference_code(fx):
    g = fx["ghmc"]
    N, D = g["N"], g["D"]
    sig = ladder(D, g["lo"], g["hi"])
    fn = otargets.diag_gaussian((f32(1) / (sig * sig)).astype(f32))
    q0 = (sig * prng.normal(prng.key(g["q0_key_seed"]), (N, D))).astype(f32)
    st = oghmc.init(q0, fn, np.asarray(g["init_key"], np.uint32))
    # (bit-equal on the stand-in, whose jax.random IS the oracle's; XLA's f32 log1p inside erf_inv is not correctly rounded,
    # so a real JAX's normal draws may sit an ulp or two away)
    np.testing.assert_allclose(st.momentum, unhex(g["init_momentum"]), rtol=5e-7, atol=1e-7)
    np.testing.assert_allclose(st.slice, unhex(g["init_slice"]), rtol=5e-7, atol=1e-7)
    for k, rec in zip(np.asarray(g["step_keys"], np.uint32), g["steps"]):
        st, info = oghmc.kernel(k, st, fn, g["eps"], sig, g["alpha"], g["delta"])
        assert info.is_accepted.astype(int).tolist() == rec["is_accepted"]
        np.testing.assert_allclose(info.acceptance_rate, unhex(rec["acceptance_rate"]), rtol=0, atol=2e-5)
        np.testing.assert_allclose(st.position, unhex(rec["position"]), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(st.momentum, unhex(rec["momentum"]), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(st.slice, unhex(rec["slice"]), rtol=2e-4, atol=2e-6)


def test_meads_equals_the_reference_code(fx):
    """``meads_adaptation(...).run``, 12 steps, 16 chains in 4 folds (fold freezing, cross-fold roll, three reshuffles)."""
    from oracle import meads as omeads

    m = fx["meads"]
    N, D = m["N"], m["D"]
    sig = ladder(D, m["lo"], m["hi"])
    fn = otargets.diag_gaussian((f32(1) / (sig * sig)).astype(f32))
    q0 = (sig * prng.normal(prng.key(m["q0_key_seed"]), (N, D))).astype(f32)
    last, params, hist = omeads.run(np.asarray(m["run_key"], np.uint32), (f32(m["q0_scale"]) * q0).astype(f32), fn,
                                    m["num_steps"], num_folds=m["num_folds"])
    assert np.stack([h[2].is_accepted for h in hist]).astype(int).tolist() == m["is_accepted_per_step"]
    np.testing.assert_allclose(np.stack([h[1].step_size for h in hist]), unhex(m["step_size_per_step"]), rtol=2e-5)
    np.testing.assert_allclose(np.stack([h[1].alpha for h in hist]), unhex(m["alpha_per_step"]), rtol=2e-5)
    np.testing.assert_allclose(np.stack([h[1].delta for h in hist]), unhex(m["delta_per_step"]), rtol=2e-5)
    np.testing.assert_allclose(last.position, unhex(m["final_position"]), rtol=1e-4, atol=5e-5)
    for name, v in m["parameters"].items():
        np.testing.assert_allclose(params[name], unhex(v), rtol=2e-5)


def _chees_setup(c):
    from oracle import chees as ochees

    N, D, T = c["N"], c["D"], c["T"]
    sig = ladder(D, c["lo"], c["hi"])
    fn = otargets.diag_gaussian((f32(1) / (sig * sig)).astype(f32))
    q0 = prng.normal(prng.key(c["q0_key_seed"]), (N, D))
    return ochees, N, D, T, fn, q0


def test_chees_run_equals_the_reference_code(fx):
    """``chees_adaptation(...).run`` as a whole (pooled statistics damp rounding differences: 40 steps stay within 1e-5):
    per-step step size, trajectory length and leapfrog counts, accept bits, final parameters.  (The optimiser on the
    reference side is tests/refshim/optax's restatement of Adam: optax itself is third party, not under /root/reference.)"""
    c = fx["chees"]
    ochees, N, D, T, fn, q0 = _chees_setup(c)
    rec = lambda t, state, info, adapt: (state.position.copy(), info, adapt)  # noqa: E731
    last, rga, params, (adapt, hist) = ochees.run(fn, prng.key(c["run_key_seed"]), q0, c["initial_step_size"],
                                                  ochees.Adam(**c["adam"]), T, num_chains=N, record=rec)
    assert [int(h[1].num_integration_steps) for h in hist] == [n[0] for n in c["num_integration_steps"]]
    assert np.stack([h[1].is_accepted for h in hist]).astype(int).tolist() == c["is_accepted"]
    assert np.stack([h[1].is_divergent for h in hist]).astype(int).tolist() == c["is_divergent"]
    np.testing.assert_allclose([h[2].step_size for h in hist], unhex(c["step_size"]), rtol=2e-5)
    np.testing.assert_allclose([h[2].trajectory_length for h in hist], unhex(c["trajectory_length"]), rtol=2e-5)
    np.testing.assert_allclose(np.stack([h[0] for h in hist]), unhex(c["position"]), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(params["step_size"], unhex(c["final_step_size"]), rtol=2e-6)
    np.testing.assert_allclose(params["integration_steps_params"][0], unhex(c["final_num_leapfrog"]), rtol=2e-5)
    assert [int(adapt.random_generator_arg)] * N == c["final_random_generator_arg"]


def test_chees_updates_equal_the_reference_code_step_by_step(fx):
    """Teacher forcing: the reference's adaptation state at t, its step-t proposals / acceptance probabilities -> the
    oracle's ``update`` (harmonic-mean acceptance, dual averaging, ChEES criterion gradient, Adam, clipping, moving
    averages) -> the reference's state at t + 1; and the transition of each step from the reference's positions."""
    c = fx["chees"]
    ochees, N, D, T, fn, q0 = _chees_setup(c)
    max_bits = int(np.ceil(np.log2(T + 1000)))  # chees_adaptation.py:761-765 (max_sampling_steps = 1000)
    jitter = lambda i: f32(f32(ochees.halton_sequence(i, max_bits) * f32(1.0)) + f32(0.0))  # noqa: E731
    init, update = ochees.base(jitter, lambda i: i + 1, ochees.Adam(**c["adam"]), ochees.OPTIMAL_TARGET_ACCEPTANCE_RATE,
                               0.5, 1000, True)
    st = init(0, c["initial_step_size"])
    pos, pp, pm = unhex(c["position"]), unhex(c["proposal_position"]), unhex(c["proposal_momentum"])
    acc, div = unhex(c["acceptance_rate"]), np.asarray(c["is_divergent"], bool)
    names = ("step_size", "log_step_size_ma", "trajectory_length", "log_trajectory_length_ma", "da_log_x", "da_log_x_avg",
             "da_avg_error", "da_mu", "adam_mu", "adam_nu")
    F = {k: unhex(c[k]) for k in names}
    keys = prng.split(prng.key(c["run_key_seed"]), T)
    for t in range(T):
        prev = q0 if t == 0 else pos[t - 1]
        L = ochees.integration_steps(jitter(st.random_generator_arg), f32(st.trajectory_length / st.step_size))
        assert L == c["num_integration_steps"][t][0], t
        s2, info = ohmc.kernel(keys[t], ohmc.init(prev, fn), fn, st.step_size, np.ones(D, f32), L)
        assert info.is_accepted.astype(int).tolist() == c["is_accepted"][t], t
        np.testing.assert_allclose(s2.position, pos[t], rtol=1e-4, atol=2e-5)
        new = update(st, pp[t], pm[t], prev, acc[t], div[t], np.ones(D, f32))
        got = {"step_size": new.step_size, "log_step_size_ma": new.log_step_size_moving_average,
               "trajectory_length": new.trajectory_length, "log_trajectory_length_ma": new.log_trajectory_length_moving_average,
               "da_log_x": new.da_state.log_step_size, "da_log_x_avg": new.da_state.log_step_size_avg,
               "da_avg_error": new.da_state.avg_error, "da_mu": new.da_state.mu,
               "adam_mu": new.optim_state.mu, "adam_nu": new.optim_state.nu}
        for k, v in got.items():
            np.testing.assert_allclose(v, F[k][t], rtol=2e-5, atol=2e-7, err_msg=f"{k} at step {t}")
        assert (new.random_generator_arg, new.step, new.da_state.step, new.optim_state.count) == (
            c["random_generator_arg"][t], c["step"][t], c["da_step"][t], c["adam_count"][t]), t
        st = ochees.ChEESAdaptationState(  # continue from the REFERENCE's state
            F["step_size"][t], F["log_step_size_ma"][t], F["trajectory_length"][t], F["log_trajectory_length_ma"][t],
            oad.DualAveragingState(F["da_log_x"][t], F["da_log_x_avg"][t], int(c["da_step"][t]), F["da_avg_error"][t], F["da_mu"][t]),
            type(new.optim_state)(int(c["adam_count"][t]), F["adam_mu"][t], F["adam_nu"][t]),
            c["random_generator_arg"][t], c["step"][t])


@pytest.mark.parametrize("name", sorted(FX["warmup"]))
def test_product_free_running_table_follows_the_reference_counters(fx, name):
    """``blackjax_amd.adaptation.free_running_table`` (the per-step scalars a free-running NUTS warm-up reads on the device:
    ``step + t0``, ``step^-kappa``, ``sqrt(step) / gamma``, the Welford count, the window-end blend coefficients) against the
    counters the REFERENCE's run went through: its dual-averaging step counter -- re-initialised at every window end -- and
    its Welford sample size."""
    from blackjax_amd import _lib
    from blackjax_amd.adaptation import free_running_table

    c = fx["warmup"][name]
    T = c["T"]
    shrink = c.get("shrinkage", 0.0)
    tab = free_running_table(T, shrink)
    AT = _lib.NUTS_AT
    da_after, wn_after = np.asarray(c["da_step"])[0], np.asarray(c["welford_n"])[0]  # chain 0: counters are shared
    sched = oad.build_schedule(T)
    wel = 0
    for t, (stage, end) in enumerate(sched):
        step_used = 1 if t == 0 else int(da_after[t - 1])  # the counter the update at t reads
        assert tab[t, AT["DA_REG"]] == f32(step_used) + f32(10.0), t
        assert tab[t, AT["DA_ETA"]] == f32(float(step_used) ** -0.75), t
        assert tab[t, AT["DA_COEF"]] == np.sqrt(f32(step_used)) / f32(0.05), t
        assert int(tab[t, AT["FLAGS"]]) == (1 if stage == 1 else 0) | (2 if end else 0), t
        if stage == 1:
            wel += 1
            assert tab[t, AT["WEL_N"]] == wel, t
        if end:
            assert int(da_after[t]) == 1 and int(wn_after[t]) == 0, t      # the reference re-initialised both
            denom = f32(wel + 5) + f32(shrink)
            assert tab[t, AT["FIN_NM1"]] == wel - 1 and tab[t, AT["FIN_BETA_DATA"]] == f32(wel) / denom, t
            assert tab[t, AT["FIN_BETA_PREV"]] == f32(shrink) / denom, t
            wel = 0
        else:
            assert int(da_after[t]) == step_used + 1, t
            assert int(wn_after[t]) == wel, t


def test_product_chees_host_update_equals_the_reference_code_step_by_step(fx):
    """The PRODUCT's host half of the ChEES update (``blackjax_amd.chees.base(...)[1].scalar_update``: harmonic-mean
    acceptance, dual averaging, the product's own Adam, clipping, moving averages, trajectory-length clamp) from the
    reference's state at t to its state at t + 1, on the CPU.  The four pooled sums the device kernels hand it
    (``bjx_chees_scalars``) are formed here from the reference's recorded proposals with the oracle's criterion."""
    import importlib

    pchees = importlib.import_module("blackjax_amd.chees")
    poptim = importlib.import_module("blackjax_amd.optim")
    from oracle import chees as ochees

    c = fx["chees"]
    N, D, T = c["N"], c["D"], c["T"]
    q0 = prng.normal(prng.key(c["q0_key_seed"]), (N, D))
    max_bits = int(np.ceil(np.log2(T + 1000)))
    jitter = lambda i: f32(f32(ochees.halton_sequence(i, max_bits) * f32(1.0)) + f32(0.0))  # noqa: E731
    a = c["adam"]
    init, update = pchees.base(jitter, lambda i: i + 1, poptim.adam(a["learning_rate"], b1=a["b1"], b2=a["b2"]),
                               ochees.OPTIMAL_TARGET_ACCEPTANCE_RATE, 0.5, 1000)
    st = init(0, c["initial_step_size"])
    pos, pp, pm = unhex(c["position"]), unhex(c["proposal_position"]), unhex(c["proposal_momentum"])
    acc, div = unhex(c["acceptance_rate"]), np.asarray(c["is_divergent"], bool)
    names = ("step_size", "log_step_size_ma", "trajectory_length", "log_trajectory_length_ma", "da_log_x", "da_log_x_avg",
             "da_avg_error", "da_mu", "adam_mu", "adam_nu")
    F = {k: unhex(c[k]) for k in names}
    for t in range(T):
        prev = q0 if t == 0 else pos[t - 1]
        nd = ~div[t]
        w = np.where(nd, acc[t], f32(0.0)).astype(f32)
        per_chain = ochees.chain_criterion(pp[t], pm[t], prev, w, np.ones(D, f32), True)
        scale = f32(f32(jitter(st.random_generator_arg)) * st.trajectory_length)
        tg = (scale * per_chain).astype(f32)
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):  # an acceptance probability of 0 gives inf here, as there
            sums = [(f32(1.0) / acc[t][nd]).astype(np.float64).sum(), float(nd.sum()),
                    (acc[t][nd].astype(np.float64) * tg[nd].astype(np.float64)).sum(),
                    (acc[t][nd] + f32(ochees.EPS_FLOAT)).astype(f32).astype(np.float64).sum()]
        new = update.scalar_update(st, sums)
        got = {"step_size": new.step_size, "log_step_size_ma": new.log_step_size_moving_average,
               "trajectory_length": new.trajectory_length, "log_trajectory_length_ma": new.log_trajectory_length_moving_average,
               "da_log_x": new.da_state.log_x, "da_log_x_avg": new.da_state.log_x_avg, "da_avg_error": new.da_state.avg_error,
               "da_mu": new.da_state.mu, "adam_mu": new.optim_state.mu, "adam_nu": new.optim_state.nu}
        for k, v in got.items():
            np.testing.assert_allclose(v, F[k][t], rtol=2e-5, atol=2e-7, err_msg=f"{k} at step {t}")
        assert (new.random_generator_arg, new.step, new.da_state.step, new.optim_state.count) == (
            c["random_generator_arg"][t], c["step"][t], c["da_step"][t], c["adam_count"][t]), t
        st = pchees.ChEESAdaptationState(  # continue from the REFERENCE's state
            F["step_size"][t], F["log_step_size_ma"][t], F["trajectory_length"][t], F["log_trajectory_length_ma"][t],
            pchees.DualAveragingState(F["da_log_x"][t], F["da_log_x_avg"][t], int(c["da_step"][t]), F["da_avg_error"][t], F["da_mu"][t]),
            poptim.ScaleByAdamState(int(c["adam_count"][t]), F["adam_mu"][t], F["adam_nu"][t]),
            c["random_generator_arg"][t], c["step"][t])


def test_c1_posterior_moments_equal_the_reference_code(fx):
    """BASELINE.json configs[0] over its 100 transitions: every one of the 12 800 accept bits equal, the
    posterior moments over all draws within 1e-6 (north_star asks for 1e-5; measured 5e-9 / 2e-8), the final positions
    within 2e-5 -- an isotropic Gaussian at eps L = 1 does not amplify rounding differences."""
    c = fx["c1_moments"]
    N, D, L, T = c["N"], c["D"], c["L"], c["T"]
    fn = otargets.diag_gaussian(np.ones(D, f32))
    q0 = prng.normal(prng.key(c["q0_key_seed"]), (N, D))
    st, pos, infos = ohmc.run(prng.key(c["run_key_seed"]), ohmc.init(q0, fn), fn, f32(c["eps"]), np.ones(D, f32), L, T)
    assert np.stack([i.is_accepted for i in infos]).astype(int).tolist() == c["is_accepted"]
    P = pos.astype(np.float64)
    assert np.abs(P.mean((0, 1)) - np.asarray(c["mean"])).max() < 1e-6
    assert np.abs(P.var((0, 1)) - np.asarray(c["var"])).max() < 1e-6
    np.testing.assert_allclose(st.position[c["rows"]], unhex(c["final_position_rows"]), rtol=0, atol=2e-5)
    rate = np.stack([i.acceptance_rate for i in infos]).astype(np.float64).mean()
    assert abs(rate - c["mean_acceptance_rate"]) < 1e-5


def test_host_helpers_equal_the_reference_code(fx):
    """The PRODUCT's host-side Halton helpers (blackjax_amd/dynamic_hmc.py) against dynamic_hmc.py:205-223 executed."""
    import importlib

    product = importlib.import_module("blackjax_amd.dynamic_hmc")
    h = fx["host_helpers"]
    for bits, ref in h["halton"].items():
        assert [product.halton_sequence(i, int(bits)) for i in range(70)] == unhex(ref).tolist(), bits
    for adj, ref in h["trajectory_length"].items():
        assert [product.halton_trajectory_length(i, float(adj)) for i in range(70)] == ref, adj
    for mu, ref in h["rescale"].items():
        np.testing.assert_allclose(product.rescale(float(mu)), unhex(ref), rtol=2e-7)


@pytest.mark.parametrize("name", sorted(FX["diagnostics"]))
def test_diagnostics_equal_the_reference_code(fx, name):
    """``effective_sample_size`` / ``rhat`` / ``potential_scale_reduction`` / ``ess_bulk`` / ``ess_tail`` of the reference
    (diagnostics.py, executed) on NumPy-generated chains: the PRODUCT's implementation on CPU tensors, and the oracle's."""
    import torch

    import blackjax_amd.diagnostics as product
    from oracle import diagnostics as odiag

    r = fx["diagnostics"][name]
    x = unhex(r["x"])
    for fn, key in (("effective_sample_size", "ess"), ("rhat", "rhat"), ("potential_scale_reduction", "psr"),
                    ("ess_bulk", "ess_bulk"), ("ess_tail", "ess_tail")):
        ref = unhex(r[key])
        np.testing.assert_allclose(np.asarray(getattr(product, fn)(torch.as_tensor(x))), ref, rtol=2e-5, err_msg=f"product {fn}")
        np.testing.assert_allclose(np.asarray(getattr(odiag, fn)(x)), ref, rtol=2e-5, err_msg=f"oracle {fn}")


@pytest.mark.skipif(not os.path.isdir("/root/reference/blackjax"), reason="/root/reference is not on this box")
def test_generator_reproduces_the_committed_fixture():
    """The committed file IS what the generator writes today (the reference on the stand-in, in its own process)."""
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, BJX_REF_SHIM_OUT=os.path.join(tmp, "out.json"), BJX_REF_SHIM_ONLY="samplers:hmc_rejections,nuts_funnel,mhmc;schedules;ghmc;diagnostics")
        subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_ref_shim_fixtures.py")], check=True, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        with open(env["BJX_REF_SHIM_OUT"]) as fh:
            again = json.load(fh)
    for name in ("hmc_rejections", "nuts_funnel", "mhmc"):
        assert again["samplers"][name] == FX["samplers"][name], name
    assert again["schedules"] == FX["schedules"] and again["ghmc"] == FX["ghmc"] and again["diagnostics"] == FX["diagnostics"]
    assert again["reference_sources_sha256"] == FX["reference_sources_sha256"]


# ------------------------------------------------------------------------------------------------ the HIP path
_GPU_CASES = [n for n, c in sorted(FX["samplers"].items()) if c["algorithm"] in ("hmc", "mhmc", "nuts")]  # (dynamic: per-chain keys)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _GPU_CASES)
def test_hip_transition_equals_the_reference_code(fx, dev, name):
    """The HIP kernels against the reference's own code, no oracle in between (the engine's shared dense metric runs on
    the fp32 MFMA GEMM: the same 1e-5 as the oracle's fp64-accumulated products)."""
    import torch

    import blackjax_amd as bjx

    c = fx["samplers"][name]
    N, D = c["N"], c["D"]
    t = c["target"]
    if t["kind"] == "diag_gaussian":
        s = ladder(D, t["lo"], t["hi"])
        fn = bjx.targets.DiagGaussian(torch.as_tensor((f32(1) / (s * s)).astype(f32), device=dev))
    elif t["kind"] == "funnel":
        fn = bjx.targets.NealFunnel()
    else:
        fn = bjx.targets.AR1Gaussian(t["rho"], D)
    imm = torch.as_tensor(metric_of(c, D), device=dev)
    integ = getattr(bjx.integrators, c.get("integrator", "velocity_verlet"))
    kw = dict(integrator=integ, divergence_threshold=c.get("divergence_threshold", 1000))
    if c["algorithm"] == "nuts":
        alg = bjx.nuts(fn, c["eps"], imm, max_num_doublings=c["max_num_doublings"], **kw)
    else:
        alg = getattr(bjx, c["algorithm"])(fn, c["eps"], imm, c["L"], **kw)
    st, info = alg.step(np.asarray(c["step_key"], np.uint32), alg.init(torch.as_tensor(initial_positions(c, N, D), device=dev)))
    n = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else x  # noqa: E731
    d = {k: n(v) for k, v in info._asdict().items() if not isinstance(v, tuple)}
    if c["algorithm"] == "nuts":
        d["leftmost_position"] = n(info.trajectory_leftmost_state.position)
        d["rightmost_position"] = n(info.trajectory_rightmost_state.position)
    else:
        d["proposal_position"] = n(info.proposal.position)
    _check_sampler(c, n(st.position), d, c["algorithm"] == "nuts")


_PLAIN_CASES = [n for n in _GPU_CASES if FX["samplers"][n]["target"]["kind"] == "diag_gaussian"
                and not np.ndim(FX["samplers"][n].get("metric_dense", 0))]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _PLAIN_CASES)
def test_hip_transition_with_a_plain_pytorch_logdensity_equals_the_reference_code(fx, dev, name):
    """Round 6: the DEFAULT user path -- the log-density written as a plain PyTorch function, as the reference's user
    writes it in jnp (`hmc.py:90-92`: value_and_grad(logdensity_fn)) -- traced on its first call into the generated
    value-and-gradient kernel, against the reference's own code on the same seeds and inputs."""
    import torch

    import blackjax_amd as bjx
    from blackjax_amd import _util

    c = fx["samplers"][name]
    N, D = c["N"], c["D"]
    s_ = ladder(D, c["target"]["lo"], c["target"]["hi"])
    inv_var = torch.as_tensor((f32(1) / (s_ * s_)).astype(f32), device=dev)
    fn = lambda q: -0.5 * (q * q * inv_var).sum(-1)  # noqa: E731
    imm = torch.as_tensor(metric_of(c, D), device=dev)
    integ = getattr(bjx.integrators, c.get("integrator", "velocity_verlet"))
    kw = dict(integrator=integ, divergence_threshold=c.get("divergence_threshold", 1000))
    if c["algorithm"] == "nuts":
        alg = bjx.nuts(fn, c["eps"], imm, max_num_doublings=c["max_num_doublings"], **kw)
    else:
        alg = getattr(bjx, c["algorithm"])(fn, c["eps"], imm, c["L"], **kw)
    st0 = alg.init(torch.as_tensor(initial_positions(c, N, D), device=dev))
    assert [type(v).__name__ for v in _util.value_and_grad(fn)._bjx_elementwise.values()] in (["DeviceTarget"], ["ElementwiseRowsTarget"])
    st, info = alg.step(np.asarray(c["step_key"], np.uint32), st0)
    n = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else x  # noqa: E731
    d = {k: n(v) for k, v in info._asdict().items() if not isinstance(v, tuple)}
    if c["algorithm"] == "nuts":
        d["leftmost_position"] = n(info.trajectory_leftmost_state.position)
        d["rightmost_position"] = n(info.trajectory_rightmost_state.position)
    else:
        d["proposal_position"] = n(info.proposal.position)
    _check_sampler(c, n(st.position), d, c["algorithm"] == "nuts")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FX["warmup"]))
def test_hip_adaptation_updates_equal_the_reference_code_step_by_step(dev, name):
    """The warm-up's device kernels (``bjx_da_init / bjx_da_update``, ``bjx_welford_update_*``, ``bjx_welford_final_*`` through
    the product's own wrappers) from the reference's state at t to its state at t + 1 -- the teacher-forced comparison of
    ``test_adaptation_updates_equal_the_reference_code_step_by_step`` with the HIP path in the oracle's place."""
    import torch

    from blackjax_amd import adaptation as ad

    c = FX["warmup"][name]
    N, D, T, diag = c["N"], c["D"], c["T"], c["diag"]
    pos, acc = unhex(c["position"]), unhex(c["acceptance_rate"])
    L, step, wn = _warmup_arrays(c)
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    target, shrink = c.get("target_acceptance_rate", 0.8), c.get("shrinkage", 0.0)
    ss, _ = ad._da_init(torch.full((N,), float(c.get("initial_step_size", 1.0)), device=dev), from_log_avg=False)
    m0 = _initial_imm(c, D)
    imm = dv(m0) if m0 is not None else (torch.ones(D, device=dev) if diag else torch.eye(D, device=dev))
    m2_0 = torch.zeros((N, D), device=dev) if diag else torch.zeros((N, D, D), device=dev)
    mm = ad.MassMatrixAdaptationState(imm, ad.WelfordAlgorithmState(torch.zeros((N, D), device=dev), m2_0, 0))
    for t, (stage, end) in enumerate(ad.build_schedule(T)):
        if stage == 1:
            mm = ad.MassMatrixAdaptationState(mm.inverse_mass_matrix, ad._welford_update(mm.wc_state, dv(pos[:, t])))
        ss, step_size = ad._da_update(ss, dv(acc[:, t]), target)
        if end:
            mm = ad._mm_final(mm, shrink)
            ss, step_size = ad._da_init(ss.log_step_size_avg, from_log_avg=True)
        got = {"log_step_size": ss.log_step_size, "log_step_size_avg": ss.log_step_size_avg, "avg_error": ss.avg_error,
               "mu": ss.mu, "step_size": step_size, "welford_mean": mm.wc_state.mean, "welford_m2": mm.wc_state.m2}
        for k, v in got.items():
            np.testing.assert_allclose(v.cpu().numpy(), L[k][:, t], rtol=5e-6, atol=1e-7, err_msg=f"{k} at step {t}")
        imm_now = mm.inverse_mass_matrix
        imm_now = imm_now.expand((N,) + tuple(imm_now.shape)) if imm_now.ndim == (1 if diag else 2) else imm_now
        np.testing.assert_allclose(imm_now.cpu().numpy(), L["inverse_mass_matrix"][:, t], rtol=5e-6, atol=1e-7, err_msg=f"imm at {t}")
        assert ss.step == int(step[0, t]) and mm.wc_state.sample_size == int(wn[0, t]), t
        # continue from the REFERENCE's state
        ss = ad.DualAveragingAdaptationState(dv(L["log_step_size"][:, t]), dv(L["log_step_size_avg"][:, t]), int(step[0, t]),
                                             dv(L["avg_error"][:, t]), dv(L["mu"][:, t]))
        mm = ad.MassMatrixAdaptationState(dv(L["inverse_mass_matrix"][:, t]),
                                          ad.WelfordAlgorithmState(dv(L["welford_mean"][:, t]), dv(L["welford_m2"][:, t]), int(wn[0, t])))


@pytest.mark.gpu
def test_hip_pooled_warmups_and_ghmc_equal_the_reference_code(dev):
    """``ghmc`` (three transitions), ``chees_adaptation`` (40 steps) and ``meads_adaptation`` (12 steps) on the HIP path
    against the reference's records -- the call conventions of tests/test_ghmc_gpu.py / test_chees_gpu.py."""
    import torch

    import blackjax_amd as bjx

    t2n = lambda x: x.detach().cpu().numpy()  # noqa: E731
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    # ---- ghmc
    g = FX["ghmc"]
    N, D = g["N"], g["D"]
    sig = ladder(D, g["lo"], g["hi"])
    inv_var = (f32(1) / (sig * sig)).astype(f32)
    q0 = (sig * prng.normal(prng.key(g["q0_key_seed"]), (N, D))).astype(f32)
    alg = bjx.ghmc(bjx.targets.DiagGaussian(dv(inv_var)), g["eps"], dv(sig), g["alpha"], g["delta"])
    st = alg.init(dv(q0), np.asarray(g["init_key"], np.uint32))
    np.testing.assert_allclose(t2n(st.momentum), unhex(g["init_momentum"]), rtol=5e-7, atol=1e-7)
    for k, rec in zip(np.asarray(g["step_keys"], np.uint32), g["steps"]):
        st, info = alg.step(k, st)
        assert t2n(info.is_accepted).astype(int).tolist() == rec["is_accepted"]
        np.testing.assert_allclose(t2n(st.position), unhex(rec["position"]), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(t2n(st.slice), unhex(rec["slice"]), rtol=2e-4, atol=2e-6)
    # ---- ChEES
    c = FX["chees"]
    N, D, T = c["N"], c["D"], c["T"]
    sig = ladder(D, c["lo"], c["hi"])
    fn = bjx.targets.DiagGaussian(dv((f32(1) / (sig * sig)).astype(f32)))
    warm = bjx.chees_adaptation(fn, N)
    a = c["adam"]
    (last, params), info = warm.run(prng.key(c["run_key_seed"]), dv(prng.normal(prng.key(c["q0_key_seed"]), (N, D))),
                                    c["initial_step_size"], bjx.optim.adam(a["learning_rate"], b1=a["b1"], b2=a["b2"]), T)
    assert t2n(info.info.num_integration_steps).reshape(T, -1)[:, 0].tolist() == [n[0] for n in c["num_integration_steps"]]
    np.testing.assert_allclose(t2n(info.adaptation_state.step_size), unhex(c["step_size"]), rtol=2e-5)
    np.testing.assert_allclose(t2n(info.adaptation_state.trajectory_length), unhex(c["trajectory_length"]), rtol=2e-5)
    np.testing.assert_allclose(t2n(info.state.position), unhex(c["position"]), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(float(params["step_size"]), unhex(c["final_step_size"]), rtol=2e-6)
    # ---- MEADS
    m = FX["meads"]
    N, D = m["N"], m["D"]
    sig = ladder(D, m["lo"], m["hi"])
    q0 = (f32(m["q0_scale"]) * (sig * prng.normal(prng.key(m["q0_key_seed"]), (N, D))).astype(f32)).astype(f32)
    warm = bjx.meads_adaptation(bjx.targets.DiagGaussian(dv((f32(1) / (sig * sig)).astype(f32))), N, num_folds=m["num_folds"])
    (st_g, par_g), info = warm.run(np.asarray(m["run_key"], np.uint32), dv(q0), m["num_steps"])
    assert t2n(info.info.is_accepted).astype(int).tolist() == m["is_accepted_per_step"]
    np.testing.assert_allclose(t2n(info.adaptation_state.step_size), unhex(m["step_size_per_step"]), rtol=2e-5)
    np.testing.assert_allclose(t2n(info.adaptation_state.alpha), unhex(m["alpha_per_step"]), rtol=2e-5)
    np.testing.assert_allclose(t2n(st_g.position), unhex(m["final_position"]), rtol=1e-4, atol=5e-5)
    for name, v in m["parameters"].items():
        np.testing.assert_allclose(t2n(par_g[name]) if hasattr(par_g[name], "cpu") else par_g[name], unhex(v), rtol=2e-5)
