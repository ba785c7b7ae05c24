#!/usr/bin/env python
"""Generate tests/golden/ref_shim_fixtures.json by EXECUTING THE REFERENCE'S OWN SOURCE (``/root/reference/blackjax``)
on the NumPy/torch stand-in for JAX in ``tests/refshim`` (read that package's docstring first: it is NOT JAX).

    python tests/golden/gen_ref_shim_fixtures.py            # run in the build container, from the repo root

Why this exists.  JAX cannot be installed where this repository is built, so ``tests/golden/gen_jax_fixtures.py`` (the
real thing: JAX + BlackJAX) has never run and every parity test compared the HIP path with the in-repo oracle -- a
restatement written by the same hands.  The reference is pure Python on JAX, though: given a stand-in for the ~60 JAX
functions its HMC / NUTS / adaptation code calls, ``import blackjax`` works and the reference's own control flow runs:
``hmc.kernel``, ``static_integration``, ``static_binomial_sampling``, ``iterative_nuts_proposal``,
``dynamic_multiplicative_expansion`` / ``dynamic_progressive_integration``, ``iterative_uturn_numpyro``,
``progressive_*_sampling``, ``gaussian_euclidean`` / ``_format_covariance``, ``generalized_two_stage_integrator``,
``dual_averaging``, ``welford_algorithm``, ``mass_matrix_adaptation``, ``staged_adaptation`` / ``build_schedule``,
``run_inference_algorithm``, ``ghmc``, ``chees_adaptation``, ``meads_adaptation``, ``diagnostics.effective_sample_size`` /
``rhat`` / ``ess_bulk`` / ``ess_tail`` -- the reference's code, unmodified, read from where it lies.

What the fixtures pin and what they do not:
* pinned: everything those functions DECIDE -- which key is split where, how a tree grows, when it stops, what is accepted,
  how the adaptation state moves -- and their arithmetic up to fp32 rounding (torch CPU kernels stand in for XLA:CPU, so
  comparisons are at a stated tolerance, as they would be against real JAX);
* NOT pinned: the ``jax.random`` bit streams.  The stand-in's ``jax.random`` IS ``oracle/prng.py``; SURVEY row a34 stays
  "parity unpinned" until ``gen_jax_fixtures.py`` runs somewhere.

tests/test_ref_shim_fixtures.py compares the oracle with this file on the CPU (and the HIP path on the GPU); the warm-up is
compared step by step from the reference's own state ("teacher forcing"): a 100-step adaptive run amplifies a one-ulp
difference by orders of magnitude whoever computes it.
"""
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # importing /root/reference/blackjax must not leave __pycache__ directories in the read-only tree

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
# BJX_REAL_JAX=1: the SAME cases on a real JAX + BlackJAX installation (any machine that has them; PYTHONPATH may point at
# a BlackJAX checkout) -> tests/golden/ref_jax_fixtures.json, which tests/test_ref_shim_fixtures.py picks up as a second
# fixture set.  That run DOES exercise jax.random and closes SURVEY row a34 for every stream these cases consume.
REAL_JAX = os.environ.get("BJX_REAL_JAX", "0") == "1"
if not REAL_JAX:
    sys.path.insert(0, os.path.join(ROOT, "tests", "refshim"))  # ONLY in this process: a module named jax lives there
    sys.path.insert(0, REF)
    # blackjax/_version.py is written by setuptools_scm at install time; the source tree has none
    _v = types.ModuleType("blackjax._version")
    _v.__version__ = "reference-source-tree"
    sys.modules["blackjax._version"] = _v
else:
    os.environ.setdefault("JAX_PLATFORMS", "cpu")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import jax  # noqa: E402  (tests/refshim/jax unless BJX_REAL_JAX=1)
import jax.numpy as jnp  # noqa: E402

assert jax.__version__.endswith("refshim") != REAL_JAX, "stand-in vs real JAX: check BJX_REAL_JAX and sys.path"
import blackjax  # noqa: E402  (/root/reference/blackjax on the stand-in)

assert REAL_JAX or os.path.realpath(blackjax.__file__).startswith(REF + "/"), blackjax.__file__


def words(k):
    return np.asarray(jax.random.key_data(k)).astype(np.uint32).tolist()


def f32hex(a):
    """float32 array -> nested list of uint32 bit patterns (exact, JSON-safe incl. inf / nan)."""
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def ints(a):
    return np.asarray(a).astype(np.int64).tolist()


# ------------------------------------------------------------------------------------------------ targets (jnp)
def sigma_ladder(D, lo, hi):
    return (10.0 ** (lo + (hi - lo) * np.arange(D) / max(D - 1, 1))).astype(np.float32)


def make_target(spec):
    kind = spec["kind"]
    if kind == "diag_gaussian":
        sig = sigma_ladder(spec["D"], spec["lo"], spec["hi"])
        inv_var = jnp.asarray((np.float32(1) / (sig * sig)).astype(np.float32))
        return lambda q: -0.5 * jnp.sum(q * q * inv_var)
    if kind == "funnel":  # tests/fixtures.py:81-98 without the normalising constants
        D = spec["D"]

        def funnel(q):
            y, v = q[0], q[1:]
            return -0.5 * (y / 3.0) ** 2 - 0.5 * jnp.exp(-y) * jnp.sum(v * v) - 0.5 * (D - 1) * y

        return funnel
    if kind == "ar1":  # Sigma_ij = rho^|i-j|: tridiagonal precision
        rho, D = spec["rho"], spec["D"]
        c = np.float32(1.0 / (1.0 - rho * rho))
        d = np.full(D, 1.0 + rho * rho, np.float32)
        d[0] = d[-1] = 1.0
        diag = jnp.asarray((d * c).astype(np.float32))
        off = float(np.float32(-rho) * c)
        return lambda q: -0.5 * (jnp.sum(diag * q * q) + 2.0 * off * jnp.sum(q[:-1] * q[1:]))
    raise ValueError(kind)


def ar1_covariance(rho, D):
    i = np.arange(D)
    return (rho ** np.abs(i[:, None] - i[None, :])).astype(np.float32)


def initial_positions(spec, N, D):
    q = jax.random.normal(jax.random.key(spec["q0_key_seed"]), (N, D), jnp.float32)
    scale = spec.get("q0_scale")
    if scale == "sigma":
        return jnp.asarray(sigma_ladder(D, spec["target"]["lo"], spec["target"]["hi"])) * q
    return q if scale is None else jnp.asarray(np.float32(scale)) * q


INTEGRATORS = {"velocity_verlet": "velocity_verlet", "mclachlan": "mclachlan", "yoshida": "yoshida", "omelyan": "omelyan"}


def metric_of(spec, D):
    m = spec["metric"]
    if m == "identity":
        return jnp.ones(D, jnp.float32)
    if m == "ladder":
        sig = sigma_ladder(D, spec["target"]["lo"], spec["target"]["hi"])
        return jnp.asarray((sig * sig).astype(np.float32))
    if m == "ar1_dense":
        return jnp.asarray(ar1_covariance(spec["metric_rho"], D))
    raise ValueError(m)


# ------------------------------------------------------------------------------------------------ sampler transitions
def sampler_case(spec):
    """One vmapped transition of ``blackjax.<algorithm>`` with ``split(step_key, N)`` chain keys."""
    import blackjax.mcmc.integrators as ref_integrators

    N, D = spec["N"], spec["D"]
    fn = make_target(dict(spec["target"], D=D))
    imm = metric_of(spec, D)
    integ = getattr(ref_integrators, INTEGRATORS[spec.get("integrator", "velocity_verlet")])
    algo = spec["algorithm"]
    kw = dict(integrator=integ, divergence_threshold=spec.get("divergence_threshold", 1000))
    if algo in ("hmc", "mhmc"):
        alg = getattr(blackjax, algo)(fn, spec["eps"], imm, spec["L"], **kw)
    elif algo == "nuts":
        alg = blackjax.nuts(fn, spec["eps"], imm, max_num_doublings=spec["max_num_doublings"], **kw)
    elif algo in ("dynamic_hmc", "dmhmc"):
        alg = getattr(blackjax, algo)(fn, spec["eps"], imm, **kw)
    else:
        raise ValueError(algo)
    q0 = initial_positions(spec, N, D)
    if algo in ("dynamic_hmc", "dmhmc"):
        arg_keys = jax.random.split(jax.random.key(spec["arg_key_seed"]), N)
        states = jax.vmap(alg.init)(q0, arg_keys)
    else:
        states = jax.vmap(alg.init)(q0)
    step_key = jax.random.key(spec["step_key_seed"])
    new, info = jax.jit(jax.vmap(alg.step))(jax.random.split(step_key, N), states)
    out = dict(spec)
    # per-chain scalars for every chain; (N, D) arrays for every chain of a small case, for the first and the last row of a large one
    rows = list(range(N)) if N * D <= 4096 else [0, N - 1]
    sel = lambda a: f32hex(np.asarray(a)[rows])  # noqa: E731
    out.update(step_key=words(step_key), rows=rows, position=sel(new.position), logdensity=f32hex(new.logdensity),
               acceptance_rate=f32hex(info.acceptance_rate), is_divergent=ints(info.is_divergent),
               energy=f32hex(info.energy), momentum=sel(info.momentum),
               num_integration_steps=ints(info.num_integration_steps))
    if algo == "nuts":
        out.update(num_trajectory_expansions=ints(info.num_trajectory_expansions), is_turning=ints(info.is_turning),
                   leftmost_position=sel(info.trajectory_leftmost_state.position),
                   rightmost_position=sel(info.trajectory_rightmost_state.position))
    else:
        out.update(is_accepted=ints(info.is_accepted), proposal_position=sel(info.proposal.position))
    if algo in ("dynamic_hmc", "dmhmc"):
        out.update(next_random_generator_arg=words(new.random_generator_arg))
    return out


SAMPLER_CASES = [
    # BASELINE.json configs[0] (the reference's CPU-runnable case), as in gen_jax_fixtures.py
    dict(name="hmc_c1", algorithm="hmc", N=128, D=1024, L=10, eps=0.1, metric="identity",
         target=dict(kind="diag_gaussian", lo=0.0, hi=0.0), q0_key_seed=1, step_key_seed=0),
    # BASELINE.json configs[1] / [4] / [2] at their own D, L, eps (fewer chains: a chain is a Python loop here)
    dict(name="hmc_c2_like", algorithm="hmc", N=16, D=1024, L=50, eps=0.25, metric="ladder",
         target=dict(kind="diag_gaussian", lo=-1.0, hi=1.0), q0_key_seed=1, q0_scale="sigma", step_key_seed=60),
    dict(name="hmc_c5_like", algorithm="hmc", N=16, D=512, L=20, eps=0.5, metric="ar1_dense", metric_rho=0.9,
         target=dict(kind="ar1", rho=0.9), q0_key_seed=1, step_key_seed=61),
    dict(name="nuts_c3_like", algorithm="nuts", N=48, D=256, eps=0.1, max_num_doublings=10, metric="identity",
         target=dict(kind="funnel"), q0_key_seed=1, step_key_seed=62),
    dict(name="hmc_ladder_ideal_mass", algorithm="hmc", N=32, D=64, L=12, eps=0.25, metric="ladder",
         target=dict(kind="diag_gaussian", lo=-1.0, hi=1.0), q0_key_seed=5, q0_scale="sigma", step_key_seed=6),
    dict(name="hmc_rejections", algorithm="hmc", N=64, D=16, L=7, eps=0.9, metric="identity",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=7, step_key_seed=8),
    dict(name="hmc_divergent", algorithm="hmc", N=16, D=8, L=8, eps=0.64, metric="identity", divergence_threshold=50,
         target=dict(kind="diag_gaussian", lo=-0.5, hi=0.5), q0_key_seed=9, step_key_seed=10),  # 14 of 16 diverge
    dict(name="hmc_all_divergent", algorithm="hmc", N=8, D=8, L=5, eps=40.0, metric="identity",
         target=dict(kind="diag_gaussian", lo=-0.5, hi=0.5), q0_key_seed=9, step_key_seed=10),  # energies overflow
    dict(name="hmc_dense_ar1", algorithm="hmc", N=32, D=16, L=8, eps=0.5, metric="ar1_dense", metric_rho=0.9,
         target=dict(kind="ar1", rho=0.9), q0_key_seed=11, step_key_seed=12),
    dict(name="hmc_mclachlan", algorithm="hmc", N=16, D=12, L=6, eps=0.4, metric="identity", integrator="mclachlan",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=13, step_key_seed=14),
    dict(name="hmc_yoshida", algorithm="hmc", N=16, D=12, L=6, eps=0.4, metric="identity", integrator="yoshida",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=13, step_key_seed=15),
    dict(name="hmc_omelyan_dense", algorithm="hmc", N=16, D=12, L=5, eps=0.5, metric="ar1_dense", metric_rho=0.7,
         integrator="omelyan", target=dict(kind="ar1", rho=0.7), q0_key_seed=13, step_key_seed=16),
    dict(name="mhmc", algorithm="mhmc", N=32, D=16, L=9, eps=0.35, metric="identity",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=17, step_key_seed=18),
    dict(name="mhmc_dense", algorithm="mhmc", N=16, D=10, L=6, eps=0.4, metric="ar1_dense", metric_rho=0.8,
         target=dict(kind="ar1", rho=0.8), q0_key_seed=19, step_key_seed=20),
    dict(name="dynamic_hmc", algorithm="dynamic_hmc", N=32, D=12, eps=0.3, metric="identity",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=21, step_key_seed=22, arg_key_seed=23),
    dict(name="dmhmc", algorithm="dmhmc", N=24, D=10, eps=0.3, metric="identity",
         target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=36, step_key_seed=37, arg_key_seed=38),
    # as in gen_jax_fixtures.py
    dict(name="nuts_funnel", algorithm="nuts", N=16, D=10, eps=0.2, max_num_doublings=6, metric="identity",
         target=dict(kind="funnel"), q0_key_seed=2, q0_scale=0.5, step_key_seed=4),
    dict(name="nuts_funnel_deep", algorithm="nuts", N=24, D=6, eps=0.04, max_num_doublings=7, metric="identity",
         target=dict(kind="funnel"), q0_key_seed=24, q0_scale=0.5, step_key_seed=25),
    dict(name="nuts_depth_limit_2", algorithm="nuts", N=16, D=8, eps=0.05, max_num_doublings=2, metric="identity",
         target=dict(kind="diag_gaussian", lo=0.0, hi=0.0), q0_key_seed=26, step_key_seed=27),
    dict(name="nuts_depth_limit_10", algorithm="nuts", N=4, D=4, eps=0.0005, max_num_doublings=10, metric="identity",
         target=dict(kind="diag_gaussian", lo=0.0, hi=0.0), q0_key_seed=58, step_key_seed=59),  # 1 023 leaves per chain
    dict(name="nuts_divergent", algorithm="nuts", N=32, D=8, eps=0.52, max_num_doublings=6, metric="identity",
         divergence_threshold=20, target=dict(kind="diag_gaussian", lo=-0.6, hi=0.6), q0_key_seed=28, q0_scale="sigma",
         step_key_seed=29),
    dict(name="nuts_ladder_ideal_mass", algorithm="nuts", N=24, D=32, eps=0.35, max_num_doublings=6, metric="ladder",
         target=dict(kind="diag_gaussian", lo=-1.0, hi=1.0), q0_key_seed=30, q0_scale="sigma", step_key_seed=31),
    dict(name="nuts_dense_ar1", algorithm="nuts", N=16, D=12, eps=0.5, max_num_doublings=5, metric="ar1_dense",
         metric_rho=0.9, target=dict(kind="ar1", rho=0.9), q0_key_seed=32, step_key_seed=33),
    dict(name="nuts_yoshida", algorithm="nuts", N=16, D=10, eps=0.5, max_num_doublings=5, metric="identity",
         integrator="yoshida", target=dict(kind="diag_gaussian", lo=-0.3, hi=0.3), q0_key_seed=34, step_key_seed=35),
]


# ------------------------------------------------------------------------------------------------ run_inference_algorithm
def run_inference_case():
    """util.py:150-213: ``run_inference_algorithm`` on ``vmap``-ped chains is not how the reference batches (one chain per
    call); the step-major layout the engine mirrors is scan-over-steps of vmap-over-chains (docs/examples/
    howto_sample_multiple_chains.md:116-130): keys[t] -> split(keys[t], N)."""
    N, D, L, T, eps = 8, 6, 4, 5, 0.3
    spec = dict(N=N, D=D, L=L, T=T, eps=eps, q0_key_seed=40, run_key_seed=41, target=dict(kind="diag_gaussian", lo=-0.2, hi=0.2))
    fn = make_target(dict(spec["target"], D=D))
    alg = blackjax.hmc(fn, eps, jnp.ones(D), L)
    q0 = jax.random.normal(jax.random.key(40), (N, D))
    states = jax.vmap(alg.init)(q0)

    def one_step(st, k):
        st, info = jax.vmap(alg.step)(jax.random.split(k, N), st)
        return st, (st.position, info.is_accepted)

    final, (positions, accepted) = jax.lax.scan(one_step, states, jax.random.split(jax.random.key(41), T))
    spec.update(positions=f32hex(positions), is_accepted=ints(accepted))
    # ... and the reference's own driver on ONE chain (initial_position path: the key is split once more, util.py:198-200)
    single_final, hist = blackjax.util.run_inference_algorithm(
        jax.random.key(42), alg, T, initial_position=q0[0], transform=lambda s, i: (s.position, i.is_accepted))
    spec.update(single_chain_key_seed=42, single_chain_positions=f32hex(hist[0]), single_chain_is_accepted=ints(hist[1]))
    return spec


# ------------------------------------------------------------------------------------------------ warm-up, step by step
def schedule_case():
    from blackjax.adaptation.staged_adaptation import build_schedule

    out = {}
    for T in (1, 5, 19, 20, 21, 40, 60, 99, 100, 101, 149, 150, 151, 200, 333, 1000):
        out[str(T)] = [[int(a), int(bool(b))] for a, b in np.asarray(build_schedule(T)).tolist()]
    return out


def warmup_case(spec):
    """``window_adaptation(algorithm).run`` per chain (vmapped chain keys), keeping EVERY step's adaptation state."""
    N, D, T = spec["N"], spec["D"], spec["T"]
    fn = make_target(dict(spec["target"], D=D))
    kw = dict(num_integration_steps=spec["L"]) if spec["algorithm"] == "hmc" else dict(max_num_doublings=spec["max_num_doublings"])
    if spec.get("initial_imm") == "ladder":  # a diagonal start that is not the identity (sigma^2 of a 10^(+-0.3) ladder)
        s0 = sigma_ladder(D, -0.3, 0.3)
        kw["initial_inverse_mass_matrix"] = jnp.asarray((s0 * s0).astype(np.float32))
    warm = blackjax.window_adaptation(getattr(blackjax, spec["algorithm"]), fn,
                                      is_mass_matrix_diagonal=spec["diag"],
                                      imm_shrinkage_to_previous=spec.get("shrinkage", 0.0),
                                      initial_step_size=spec.get("initial_step_size", 1.0),
                                      target_acceptance_rate=spec.get("target_acceptance_rate", 0.8), **kw)
    q0 = initial_positions(spec, N, D)
    run_key = jax.random.key(spec["run_key_seed"])

    def one(k, q):
        (state, params), info = warm.run(k, q, T)
        return state.position, params["step_size"], params["inverse_mass_matrix"], info

    pos, eps, imm, info = jax.vmap(one)(jax.random.split(run_key, N), q0)
    ad = info.adaptation_state
    out = dict(spec)
    out.update(run_key=words(run_key), final_position=f32hex(pos), final_step_size=f32hex(eps), final_inverse_mass_matrix=f32hex(imm),
               position=f32hex(info.state.position), acceptance_rate=f32hex(info.info.acceptance_rate),
               log_step_size=f32hex(ad.ss_state.log_step_size), log_step_size_avg=f32hex(ad.ss_state.log_step_size_avg),
               da_step=ints(ad.ss_state.step), avg_error=f32hex(ad.ss_state.avg_error), mu=f32hex(ad.ss_state.mu),
               step_size=f32hex(ad.step_size), inverse_mass_matrix=f32hex(ad.inverse_mass_matrix),
               welford_mean=f32hex(ad.imm_state.wc_state.mean), welford_m2=f32hex(ad.imm_state.wc_state.m2),
               welford_n=ints(ad.imm_state.wc_state.sample_size))
    return out


WARMUP_CASES = [
    dict(name="warmup_hmc_diag_200", algorithm="hmc", N=2, D=6, L=5, T=200, diag=True, run_key_seed=19, q0_key_seed=3,
         q0_scale="sigma", target=dict(kind="diag_gaussian", lo=-0.4, hi=0.4)),
    dict(name="warmup_hmc_dense_100", algorithm="hmc", N=2, D=5, L=5, T=100, diag=False, run_key_seed=50, q0_key_seed=51,
         target=dict(kind="ar1", rho=0.6)),
    dict(name="warmup_hmc_diag_shrinkage", algorithm="hmc", N=2, D=6, L=4, T=100, diag=True, shrinkage=2.0,
         target_acceptance_rate=0.65, run_key_seed=52, q0_key_seed=53, q0_scale="sigma",
         target=dict(kind="diag_gaussian", lo=-0.4, hi=0.4)),
    dict(name="warmup_hmc_diag_given_start", algorithm="hmc", N=2, D=6, L=4, T=60, diag=True, initial_step_size=0.3,
         initial_imm="ladder", run_key_seed=56, q0_key_seed=57, q0_scale="sigma", target=dict(kind="diag_gaussian", lo=-0.4, hi=0.4)),
    dict(name="warmup_nuts_diag_100", algorithm="nuts", N=2, D=6, max_num_doublings=5, T=100, diag=True, run_key_seed=54,
         q0_key_seed=55, q0_scale="sigma", target=dict(kind="diag_gaussian", lo=-0.4, hi=0.4)),
]


# ------------------------------------------------------------------------------------------------ ghmc
def ghmc_case():
    N, D = 16, 6
    sig = sigma_ladder(D, -0.5, 0.5)
    fn = make_target(dict(kind="diag_gaussian", lo=-0.5, hi=0.5, D=D))
    q0 = jnp.asarray(sig) * jax.random.normal(jax.random.key(21), (N, D))
    alg = blackjax.ghmc(fn, 0.7, jnp.asarray(sig), 0.4, 0.2)
    init_key = jax.random.key(7)
    states = jax.vmap(alg.init)(q0, jax.random.split(init_key, N))
    out = dict(N=N, D=D, lo=-0.5, hi=0.5, eps=0.7, alpha=0.4, delta=0.2, q0_key_seed=21, init_key=words(init_key),
               init_momentum=f32hex(states.momentum), init_slice=f32hex(states.slice), steps=[])
    step_keys = jax.random.split(jax.random.key(9), 3)
    out["step_keys"] = words(step_keys)
    for k in step_keys:
        states, info = jax.jit(jax.vmap(alg.step))(jax.random.split(k, N), states)
        out["steps"].append(dict(is_accepted=ints(info.is_accepted), acceptance_rate=f32hex(info.acceptance_rate),
                                 position=f32hex(states.position), momentum=f32hex(states.momentum),
                                 slice=f32hex(states.slice)))
    return out


# ------------------------------------------------------------------------------------------------ ChEES, MEADS, diagnostics
def chees_case():
    """``chees_adaptation(...).run`` with the optimiser of the reference's own test (tests/adaptation/test_adaptation.py:
    104: adam(0.5, b1=0, b2=0.95) -- here tests/refshim/optax's restatement of Adam), every step's transition and state."""
    import optax

    N, D, T = 16, 4, 40
    spec = dict(N=N, D=D, T=T, lo=-0.3, hi=0.3, q0_key_seed=7, run_key_seed=11, initial_step_size=0.1,
                adam=dict(learning_rate=0.5, b1=0.0, b2=0.95))
    fn = make_target(dict(kind="diag_gaussian", lo=-0.3, hi=0.3, D=D))
    q0 = jax.random.normal(jax.random.key(7), (N, D))
    warm = blackjax.chees_adaptation(fn, num_chains=N)
    (last, params), info = warm.run(jax.random.key(11), q0, step_size=0.1, optim=optax.adam(**spec["adam"]), num_steps=T)
    ad, tr = info.adaptation_state, info.info
    spec.update(
        position=f32hex(info.state.position), proposal_position=f32hex(tr.proposal.position),
        proposal_momentum=f32hex(tr.proposal.momentum), acceptance_rate=f32hex(tr.acceptance_rate),
        is_accepted=ints(tr.is_accepted), is_divergent=ints(tr.is_divergent), num_integration_steps=ints(tr.num_integration_steps),
        step_size=f32hex(ad.step_size), log_step_size_ma=f32hex(ad.log_step_size_moving_average),
        trajectory_length=f32hex(ad.trajectory_length), log_trajectory_length_ma=f32hex(ad.log_trajectory_length_moving_average),
        da_log_x=f32hex(ad.da_state.log_x), da_log_x_avg=f32hex(ad.da_state.log_x_avg), da_step=ints(ad.da_state.step),
        da_avg_error=f32hex(ad.da_state.avg_error), da_mu=f32hex(ad.da_state.mu),
        adam_count=ints(ad.optim_state.count), adam_mu=f32hex(ad.optim_state.mu), adam_nu=f32hex(ad.optim_state.nu),
        random_generator_arg=ints(ad.random_generator_arg), step=ints(ad.step),
        final_step_size=f32hex(params["step_size"]), final_num_leapfrog=f32hex(params["integration_steps_params"][0]),
        final_position=f32hex(last.position), final_random_generator_arg=ints(last.random_generator_arg))
    return spec


def meads_case():
    N, D = 16, 6
    sig = sigma_ladder(D, -0.5, 0.5)
    fn = make_target(dict(kind="diag_gaussian", lo=-0.5, hi=0.5, D=D))
    q0 = jnp.asarray(sig) * jax.random.normal(jax.random.key(21), (N, D))
    warm = blackjax.meads_adaptation(fn, num_chains=N, num_folds=4)
    run_key = jax.random.key(5)
    (last, params), info = warm.run(run_key, 1.5 * q0, num_steps=12)
    return dict(N=N, D=D, lo=-0.5, hi=0.5, q0_key_seed=21, q0_scale=1.5, num_steps=12, num_folds=4, run_key=words(run_key),
                step_size_per_step=f32hex(info.adaptation_state.step_size), alpha_per_step=f32hex(info.adaptation_state.alpha),
                delta_per_step=f32hex(info.adaptation_state.delta),
                is_accepted_per_step=ints(info.info.is_accepted), final_position=f32hex(last.position),
                parameters={k: f32hex(v) for k, v in params.items()})


def diagnostics_case():
    """``effective_sample_size`` / ``rhat`` / ``ess_bulk`` / ``ess_tail`` on NumPy-generated chains (no jax.random)."""
    rng = np.random.default_rng(12345)
    iid = rng.standard_normal((4, 200, 3)).astype(np.float32)
    e = rng.standard_normal((8, 500)).astype(np.float32)
    ar = np.zeros_like(e)
    for t in range(1, 500):
        ar[:, t] = np.float32(0.7) * ar[:, t - 1] + e[:, t]
    shifted = (rng.standard_normal((4, 300)) + np.array([0.0, 0.5, -0.5, 1.0])[:, None]).astype(np.float32)
    out = {}
    for name, x in (("iid_4x200x3", iid), ("ar1_8x500", ar), ("shifted_4x300", shifted)):
        xa = jnp.asarray(x)
        rec = dict(x=f32hex(x), ess=f32hex(blackjax.diagnostics.effective_sample_size(xa)),
                   rhat=f32hex(blackjax.diagnostics.rhat(xa)), psr=f32hex(blackjax.diagnostics.potential_scale_reduction(xa)),
                   ess_bulk=f32hex(blackjax.diagnostics.ess_bulk(xa)), ess_tail=f32hex(blackjax.diagnostics.ess_tail(xa)))
        out[name] = rec
    return out


def c1_moments_case():
    """BASELINE.json configs[0] as SURVEY.md 8(d) spells it out: blackjax.hmc on the 1 024-dim isotropic Gaussian, 128 chains,
    L = 10, eps = 0.1, T = 100 transitions (scan over split(key(0), T) of vmap over split(keys[t], N)), q0 = normal(key(1)).
    Records every accept bit and the posterior moments over all 12 800 draws (north_star: "posterior moments within 1e-5 of
    reference")."""
    N, D, L, T, eps = 128, 1024, 10, 100, 0.1
    fn = make_target(dict(kind="diag_gaussian", lo=0.0, hi=0.0, D=D))
    alg = blackjax.hmc(fn, eps, jnp.ones(D), L)
    states = jax.vmap(alg.init)(jax.random.normal(jax.random.key(1), (N, D)))

    def one_step(st, k):
        st, info = jax.vmap(alg.step)(jax.random.split(k, N), st)
        return st, (st.position, info.is_accepted, info.acceptance_rate)

    final, (pos, acc, rate) = jax.lax.scan(one_step, states, jax.random.split(jax.random.key(0), T))
    P = np.asarray(pos).astype(np.float64)
    return dict(N=N, D=D, L=L, T=T, eps=eps, q0_key_seed=1, run_key_seed=0, is_accepted=ints(acc),
                mean=np.asarray(P.mean((0, 1))).tolist(), var=np.asarray(P.var((0, 1))).tolist(),
                mean_acceptance_rate=float(np.asarray(rate).astype(np.float64).mean()), rows=[0, N - 1],
                final_position_rows=f32hex(np.asarray(final.position)[[0, N - 1]]))


def host_helpers_case():
    """dynamic_hmc.py:205-223 / adjusted_mclmc.py:281-288: the Halton helpers the engine mirrors on the host."""
    from blackjax.mcmc.adjusted_mclmc import rescale
    from blackjax.mcmc.dynamic_hmc import halton_sequence, halton_trajectory_length

    out = {"halton": {}, "trajectory_length": {}, "rescale": {}}
    for bits in (5, 10, 11):
        out["halton"][str(bits)] = f32hex([halton_sequence(jnp.asarray(i, jnp.int32), bits) for i in range(70)])
    for adj in (1.7, 5.0, 12.3):
        out["trajectory_length"][str(adj)] = ints([halton_trajectory_length(jnp.asarray(i, jnp.int32), adj) for i in range(70)])
    for mu in (1.0, 1.7, 2.5, 5.0, 12.3, 100.0):
        out["rescale"][str(mu)] = f32hex(rescale(jnp.asarray(mu)))
    return out


def sha256_of(paths):
    out = {}
    base = REF if not REAL_JAX else os.path.dirname(os.path.dirname(os.path.abspath(blackjax.__file__)))
    for p in paths:
        full = os.path.join(base, p)
        if os.path.exists(full):
            with open(full, "rb") as fh:
                out[p] = hashlib.sha256(fh.read()).hexdigest()
    return out


def main():
    out = {
        "generator": "tests/golden/gen_ref_shim_fixtures.py",
        "what": ("the reference's own source executed on tests/refshim (torch/NumPy stand-in for JAX; jax.random = oracle/prng.py). "
                 "NOT produced by JAX; does not pin the jax.random bit streams (SURVEY a34).") if not REAL_JAX else
                f"BlackJAX {getattr(blackjax, '__version__', '?')} on JAX {jax.__version__} (CPU): the real reference, jax.random included.",
        "real_jax": REAL_JAX,
        "reference_sources_sha256": sha256_of([
            "blackjax/mcmc/hmc.py", "blackjax/mcmc/nuts.py", "blackjax/mcmc/trajectory.py", "blackjax/mcmc/termination.py",
            "blackjax/mcmc/proposal.py", "blackjax/mcmc/integrators.py", "blackjax/mcmc/metrics.py", "blackjax/mcmc/dynamic_hmc.py",
            "blackjax/mcmc/ghmc.py", "blackjax/util.py", "blackjax/optimizers/dual_averaging.py",
            "blackjax/adaptation/step_size.py", "blackjax/adaptation/mass_matrix.py", "blackjax/adaptation/staged_adaptation.py",
            "blackjax/adaptation/window_adaptation.py", "blackjax/adaptation/chees_adaptation.py",
            "blackjax/adaptation/meads_adaptation.py", "blackjax/diagnostics.py", "pyproject.toml"]),
        "samplers": {}, "warmup": {},
    }
    # BJX_REF_SHIM_ONLY="samplers:a,b;schedules;ghmc" regenerates a subset (tests/test_ref_shim_fixtures.py re-runs three
    # quick cases to check that the committed file is what this script writes); BJX_REF_SHIM_OUT redirects the output
    only = os.environ.get("BJX_REF_SHIM_ONLY")
    want = None
    if only:
        want = {}
        for part in only.split(";"):
            sec, _, names = part.partition(":")
            want[sec] = set(names.split(",")) if names else None

    def selected(section, name=None):
        if want is None:
            return True
        if section not in want:
            return False
        return want[section] is None or name is None or name in want[section]

    for spec in SAMPLER_CASES:
        if selected("samplers", spec["name"]):
            out["samplers"][spec["name"]] = sampler_case(spec)
            print("sampler", spec["name"], file=sys.stderr)
    if selected("run_inference"):
        out["run_inference"] = run_inference_case()
    if selected("schedules"):
        out["schedules"] = schedule_case()
    for spec in WARMUP_CASES:
        if selected("warmup", spec["name"]):
            out["warmup"][spec["name"]] = warmup_case(spec)
            print("warmup", spec["name"], file=sys.stderr)
    if selected("ghmc"):
        out["ghmc"] = ghmc_case()
    if selected("chees"):
        out["chees"] = chees_case()
    if selected("meads"):
        out["meads"] = meads_case()
    if selected("diagnostics"):
        out["diagnostics"] = diagnostics_case()
    if selected("host_helpers"):
        out["host_helpers"] = host_helpers_case()
    if selected("c1_moments"):
        out["c1_moments"] = c1_moments_case()  # ~2 minutes of Python loops
    path = os.environ.get("BJX_REF_SHIM_OUT") or os.path.join(HERE, "ref_jax_fixtures.json" if REAL_JAX else "ref_shim_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
