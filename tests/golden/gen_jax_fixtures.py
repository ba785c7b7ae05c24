#!/usr/bin/env python
"""Generate tests/golden/jax_fixtures.json from the REAL reference: JAX + BlackJAX on CPU.

Cannot run in the build container (no jax wheel, no network) -- run it on any
machine that has them, from the repo root, and commit the JSON:

    JAX_PLATFORMS=cpu PYTHONPATH=/path/to/blackjax python tests/golden/gen_jax_fixtures.py

What it records (all under jax's default ``jax_threefry_partitionable``, float32):
  prng      raw key words of key(seed), split, fold_in; bits / uniform / normal / bernoulli / randint
            draws -- the bit streams behind blackjax/util.py:90, mcmc/proposal.py:226,
            mcmc/trajectory.py:321,645-650 that the oracle (oracle/prng.py) and the kernels
            (blackjax_amd/csrc/bjx_device.h) restate from the published algorithm
  hmc_c1    one ``blackjax.hmc`` transition at BASELINE.json configs[0] (128 chains x 1 024 dims,
            L = 10, eps = 0.1, identity mass), vmapped with ``split(step_key, N)`` chain keys
  nuts_funnel  one ``blackjax.nuts`` transition, 16 chains on the 10-dim funnel (eps 0.2, depth <= 6)
  window_adaptation  a 40-step ``window_adaptation(hmc, L = 6)`` of 4 vmapped chains, 8 dims
  ghmc_meads  ``jax.random.permutation`` / ``uniform(minval=-1, maxval=1)`` streams, three vmapped
            ``blackjax.ghmc`` transitions (16 chains x 6 dims) and a 12-step ``meads_adaptation`` run
            (16 chains, 4 folds: freezing, cross-fold roll, three reshuffles)

tests/test_jax_fixtures.py loads the file when present and compares the oracle (CPU) and the HIP
path (GPU) with it; until then the RNG bit stream stays "parity unpinned" (NOTEBOOK.md section 3).

Run ALSO ``BJX_REAL_JAX=1 python tests/golden/gen_ref_shim_fixtures.py`` on that machine: the richer case set (20 sampler
cases, four warm-ups with every step's state, ChEES, MEADS, diagnostics, C1 over 100 transitions) that round 5 generated
from the reference's source on a stand-in for JAX; tests/test_ref_shim_fixtures.py picks the real-JAX file up as a second
fixture set.  (Its warm-up comparison is step by step from the reference's state: a whole adaptive run, as recorded by
``window_adaptation()`` below, amplifies one-ulp differences tenfold every few steps and will not hold 1e-3 over 40 steps.)
"""
import json
import os
import sys

import numpy as np

os.environ.setdefault("JAX_PLATFORMS", "cpu")
import jax  # noqa: E402
import jax.numpy as jnp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def words(k):
    return np.asarray(jax.random.key_data(k)).astype(np.uint32).tolist()


def f32hex(a):
    """float32 array -> nested list of uint32 bit patterns (exact, JSON-safe incl. inf/nan)."""
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def prng_fixtures():
    out = {"jax_version": jax.__version__,
           "threefry_partitionable": bool(jax.config.jax_threefry_partitionable), "cases": []}
    for seed in (0, 1, 42, 2024, (7 << 32) + 5):
        k = jax.random.key(seed)
        case = {
            "seed": seed, "key": words(k),
            "split2": words(jax.random.split(k, 2)), "split3": words(jax.random.split(k, 3)),
            "split5": words(jax.random.split(k, 5)),
            "fold_in": {str(d): words(jax.random.fold_in(k, d)) for d in (0, 1, 7, 1023)},
            "bits_7": np.asarray(jax.random.bits(k, (7,), dtype=jnp.uint32)).tolist(),
            "bits_2x3": np.asarray(jax.random.bits(k, (2, 3), dtype=jnp.uint32)).tolist(),
            "uniform_scalar": f32hex(jax.random.uniform(k, (), jnp.float32)),
            "uniform_5": f32hex(jax.random.uniform(k, (5,), jnp.float32)),
            "normal_scalar": f32hex(jax.random.normal(k, (), jnp.float32)),
            "normal_1024": f32hex(jax.random.normal(k, (1024,), jnp.float32)),
            "bernoulli_half": bool(jax.random.bernoulli(k)),
            "bernoulli_p": [bool(jax.random.bernoulli(k, p)) for p in (0.1, 0.5, 0.9)],
            "randint_1_10": int(jax.random.randint(k, (), 1, 10)),
        }
        out["cases"].append(case)
    return out


def hmc_c1():
    import blackjax

    N, D, L, eps = 128, 1024, 10, 0.1
    inv_var = jnp.ones(D, jnp.float32)

    def logdensity(q):
        return -0.5 * jnp.sum(q * q * inv_var)

    alg = blackjax.hmc(logdensity, eps, jnp.ones(D, jnp.float32), L)
    q0 = jax.random.normal(jax.random.key(1), (N, D), jnp.float32)
    states = jax.vmap(alg.init)(q0)
    step_key = jax.random.split(jax.random.key(0), 3)[0]
    keys = jax.random.split(step_key, N)
    new, info = jax.jit(jax.vmap(alg.step))(keys, states)
    rows = [0, 1, 63, 127]
    return {"blackjax_version": getattr(blackjax, "__version__", "?"),
            "N": N, "D": D, "L": L, "eps": eps, "q0_key_seed": 1, "step_key": words(step_key),
            "is_accepted": np.asarray(info.is_accepted).astype(int).tolist(),
            "is_divergent": np.asarray(info.is_divergent).astype(int).tolist(),
            "acceptance_rate": f32hex(info.acceptance_rate), "energy": f32hex(info.energy),
            "rows": rows, "momentum_rows": f32hex(np.asarray(info.momentum)[rows]),
            "position_rows": f32hex(np.asarray(new.position)[rows]),
            "proposal_position_rows": f32hex(np.asarray(info.proposal.position)[rows])}


def nuts_funnel():
    import blackjax

    N, D, eps, depth = 16, 10, 0.2, 6

    def logdensity(q):  # tests/fixtures.py:81-98 without the normalising constants
        y, v = q[0], q[1:]
        return -0.5 * (y / 3.0) ** 2 - 0.5 * jnp.exp(-y) * jnp.sum(v * v) - 0.5 * (D - 1) * y

    alg = blackjax.nuts(logdensity, eps, jnp.ones(D, jnp.float32), max_num_doublings=depth)
    q0 = 0.5 * jax.random.normal(jax.random.key(2), (N, D), jnp.float32)
    states = jax.vmap(alg.init)(q0)
    step_key = jax.random.key(4)
    new, info = jax.jit(jax.vmap(alg.step))(jax.random.split(step_key, N), states)
    return {"N": N, "D": D, "eps": eps, "max_num_doublings": depth, "q0_key_seed": 2, "q0_scale": 0.5,
            "step_key": words(step_key),
            "num_integration_steps": np.asarray(info.num_integration_steps).tolist(),
            "num_trajectory_expansions": np.asarray(info.num_trajectory_expansions).tolist(),
            "is_turning": np.asarray(info.is_turning).astype(int).tolist(),
            "is_divergent": np.asarray(info.is_divergent).astype(int).tolist(),
            "acceptance_rate": f32hex(info.acceptance_rate), "position": f32hex(new.position)}


def window_adaptation():
    import blackjax

    N, D, L, T = 4, 8, 6, 40
    sig = 10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))
    inv_var = jnp.asarray(1.0 / (sig * sig), jnp.float32)

    def logdensity(q):
        return -0.5 * jnp.sum(q * q * inv_var)

    warm = blackjax.window_adaptation(blackjax.hmc, logdensity, num_integration_steps=L)
    q0 = jnp.asarray(sig, jnp.float32) * jax.random.normal(jax.random.key(3), (N, D), jnp.float32)
    run_key = jax.random.key(19)

    def one(k, q):
        (state, params), info = warm.run(k, q, T)
        return state.position, params["step_size"], params["inverse_mass_matrix"], info.info.acceptance_rate

    pos, eps, imm, acc = jax.vmap(one)(jax.random.split(run_key, N), q0)
    return {"N": N, "D": D, "L": L, "num_steps": T, "q0_key_seed": 3, "run_key": words(run_key),
            "position": f32hex(pos), "step_size": f32hex(eps), "inverse_mass_matrix": f32hex(imm),
            "acceptance_rate_per_step": f32hex(acc)}


def ghmc_meads():
    import blackjax

    N, D = 16, 6
    sig = 10.0 ** (-0.5 + 1.0 * np.arange(D) / (D - 1))
    inv_var = jnp.asarray(1.0 / (sig * sig), jnp.float32)

    def logdensity(q):
        return -0.5 * jnp.sum(q * q * inv_var)

    out = {"N": N, "D": D,
           "permutation": {str(n): np.asarray(jax.random.permutation(jax.random.key(5), n)).tolist()
                           for n in (1, 2, 16, 1000)},
           "uniform_pm1": f32hex(jax.random.uniform(jax.random.key(6), (5,), jnp.float32, -1.0, 1.0))}
    q0 = jnp.asarray(sig, jnp.float32) * jax.random.normal(jax.random.key(21), (N, D), jnp.float32)
    alg = blackjax.ghmc(logdensity, 0.7, jnp.asarray(sig, jnp.float32), 0.4, 0.2)
    init_key = jax.random.key(7)
    states = jax.vmap(alg.init)(q0, jax.random.split(init_key, N))
    out["init_key"] = words(init_key)
    out["init_momentum"], out["init_slice"] = f32hex(states.momentum), f32hex(states.slice)
    step_keys = jax.random.split(jax.random.key(9), 3)
    out["step_keys"] = words(step_keys)
    out["steps"] = []
    for k in step_keys:
        states, info = jax.jit(jax.vmap(alg.step))(jax.random.split(k, N), states)
        out["steps"].append({"is_accepted": np.asarray(info.is_accepted).astype(int).tolist(),
                             "acceptance_rate": f32hex(info.acceptance_rate), "position": f32hex(states.position),
                             "momentum": f32hex(states.momentum), "slice": f32hex(states.slice)})
    warm = blackjax.meads_adaptation(logdensity, num_chains=N, num_folds=4)
    run_key = jax.random.key(5)
    (last, params), info = warm.run(run_key, 1.5 * q0, num_steps=12)
    out["meads"] = {"run_key": words(run_key), "num_steps": 12, "q0_scale": 1.5,
                    "step_size_per_step": f32hex(info.adaptation_state.step_size),
                    "alpha_per_step": f32hex(info.adaptation_state.alpha),
                    "is_accepted_per_step": np.asarray(info.info.is_accepted).astype(int).tolist(),
                    "final_position": f32hex(last.position),
                    "parameters": {k: f32hex(v) for k, v in params.items()}}
    return out


def main():
    out = {"generator": "tests/golden/gen_jax_fixtures.py", "prng": prng_fixtures()}
    try:
        out["hmc_c1"] = hmc_c1()
        out["nuts_funnel"] = nuts_funnel()
        out["window_adaptation"] = window_adaptation()
        out["ghmc_meads"] = ghmc_meads()
    except ImportError as e:
        print(f"blackjax not importable ({e}): wrote the prng section only", file=sys.stderr)
    path = os.path.join(HERE, "jax_fixtures.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
