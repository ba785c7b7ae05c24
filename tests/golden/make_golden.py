"""Writes tests/golden/reference_kats.json.

The reference is a JAX library and JAX cannot be imported in the build container (no
wheel, no network, Python 3.10 < 3.11), so these vectors are NOT produced by running the
reference: they are the literal golden values / truth tables held by the reference's OWN
tests, transcribed with their source location.  The only external vectors are the
Random123 known-answer tests for threefry2x32 (the block function under jax.random).
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

kats = {
    "_provenance": "literal values from /root/reference/tests (file:line per entry); see make_golden.py",
    # tests/mcmc/test_integrators.py:74-103,136-145 -- dense 6-d Gaussian, 16 velocity-Verlet
    # steps, eps = 0.005, inverse mass matrix = cov, logdensity = mvn.logpdf(q, 0, cov)
    "velocity_verlet_mvnormal": {
        "source": "tests/mcmc/test_integrators.py:74-103,136-145",
        "q_init": [0.0, 1.0, 2.0, 3.0, 1.0, 1.0],
        "p_init": [0.53288144, 0.25310317, 1.3788314, -0.13486017, -0.59082425, 1.2088736],
        "cov": [
            [5.9959664, 1.1494889, -1.0420643, -0.6328479, -0.20363973, 2.1600752],
            [1.1494889, 1.3504763, -0.3601517, -0.98311526, 1.1569028, -1.4185406],
            [-1.0420643, -0.3601517, 6.3011055, -2.0662997, -0.10126236, 1.2898219],
            [-0.6328479, -0.98311526, -2.0662997, 4.82699, -2.575554, 2.5724294],
            [-0.20363973, 1.1569028, -0.10126236, -2.575554, 3.35319, -2.9411654],
            [2.1600752, -1.4185406, 1.2898219, 2.5724294, -2.9411654, 6.3740206],
        ],
        "num_steps": 16,
        "step_size": 0.005,
        "q_final": [0.38887993, 0.85231394, 2.7879136, 3.0339851, 0.5856687, 1.9291426],
        "p_final": [0.46576163, 0.23854092, 1.2518811, -0.35647452, -0.742138, 1.2552949],
        "atol": 1e-6,
    },
    # tests/mcmc/test_integrators.py:105-135 -- analytic end points (checked there with atol 1e-2
    # plus energy conservation to 1e-4)
    "velocity_verlet_analytic": {
        "source": "tests/mcmc/test_integrators.py:105-135,173-223",
        "free_fall": {"num_steps": 100, "step_size": 0.01, "q_init": [0.0], "p_init": [1.0],
                      "q_final": [0.5], "p_final": [1.0], "imm": [1.0]},
        "harmonic_oscillator": {"num_steps": 100, "step_size": 0.01, "q_init": [0.0], "p_init": [1.0],
                                "q_final": [0.8414709848078965], "p_final": [0.5403023058681398],
                                "imm": [1.0]},
        "planetary_motion": {"num_steps": 628, "step_size": 0.01, "q_init": [1.0, 0.0],
                             "p_init": [0.0, 1.0], "q_final": [1.0, 0.0], "p_final": [0.0, 1.0],
                             "imm": [1.0, 1.0]},
        "position_atol": 1e-2, "energy_atol": 1e-4,
    },
    # tests/mcmc/test_uturn.py:11-43
    "iterative_uturn": {
        "source": "tests/mcmc/test_uturn.py:11-43",
        "momentum": 1.0, "momentum_sum": 3.0,
        "momentum_ckpts": [1.0, 2.0, 3.0, -2.0], "momentum_sum_ckpts": [2.0, 4.0, 4.0, -1.0],
        "cases": [[[3, 2], False], [[3, 3], True], [[0, 0], False], [[0, 1], True], [[1, 3], True]],
    },
    # tests/mcmc/test_trajectory.py:193-260 -- 1-d standard normal, position 0, momentum =
    # normal(key(0), (1,)), expansion key = key(0), imm = [1.], divergence_threshold = 1000:
    # (step_size, should_diverge, should_turn, expected_doublings)
    "dynamic_expansion": {
        "source": "tests/mcmc/test_trajectory.py:193-260",
        "key_seed": 0, "max_doublings": 10, "divergence_threshold": 1000,
        "cases": [[1e-10, False, False, 10], [1.0, False, True, 2], [100000.0, True, True, 1]],
    },
    # tests/adaptation/test_adaptation.py:27-49 -- run-length encoded (stage, is_window_end, count)
    "build_schedule": {
        "source": "tests/adaptation/test_adaptation.py:27-49",
        "19": [[0, False, 19]],
        "100": [[0, False, 15], [1, False, 74], [1, True, 1], [0, False, 10]],
        "200": [[0, False, 75], [1, False, 24], [1, True, 1], [1, False, 49], [1, True, 1], [0, False, 50]],
    },
    # tests/optimizers/test_optimizers.py:27-49 -- dual averaging gamma=0.3 on f=(x-1)^2 from x=3
    "dual_averaging": {
        "source": "tests/optimizers/test_optimizers.py:27-49",
        "gamma": 0.3, "x_init": 3.0, "num_updates": 100, "expected_final": 1.0, "delta": 0.1,
    },
    # tests/adaptation/test_mass_matrix.py:12-44 -- Welford recovers cov of np.random.seed(0) samples
    "welford": {
        "source": "tests/adaptation/test_mass_matrix.py:12-44",
        "numpy_seed": 0, "num_samples": 3000, "rtol": 0.1,
    },
    # Random123 KATs for threefry2x32 (20 rounds): key, counter -> output
    "threefry2x32": {
        "source": "Random123 kat_vectors (external; jax/_src/prng.py threefry2x32)",
        "cases": [
            [[0, 0], [0, 0], [0x6B200159, 0x99BA4EFE]],
            [[0xFFFFFFFF, 0xFFFFFFFF], [0xFFFFFFFF, 0xFFFFFFFF], [0x1CB996FC, 0xBB002BE7]],
            [[0x13198A2E, 0x03707344], [0x243F6A88, 0x85A308D3], [0xC4923A9C, 0x483DF7A0]],
        ],
    },
    # Values printed in JAX's own public documentation (jax.random tutorial) for the
    # partitionable threefry layout (transcribed from memory of the docs; flagged "unpinned"
    # in DESIGN.md because JAX cannot be run here to confirm them):
    "jax_docs_streams": {
        "source": "JAX docs 'Pseudorandom numbers' (jax >= 0.5 defaults), unverified here",
        "split_key0": [[1797259609, 2579123966], [928981903, 3453687069]],
        "normal_key42_scalar": -0.028304616,
        "legacy_split_key0_words": [4146024105, 2718843009],
        # three more values printed in JAX's documentation (jax.random module docs / "Pseudorandom numbers"
        # tutorial, partitionable layout), supplied by the round-3 review; same status: doc-sourced,
        # not reproducible here
        "uniform_key0_scalar": 0.947667,
        "split_key42": [[1832780943, 270669613], [64467757, 2916123636]],
        "normal_key0_3": [1.6226422, 2.0252647, -0.43359444],
    },
}

with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
    json.dump(kats, f, indent=1)
print("wrote reference_kats.json")
