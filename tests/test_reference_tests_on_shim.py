"""The reference's OWN unit tests for the hot path, run on the stand-in for JAX (``tests/refshim``): U-turn truth table,
trajectory integration (divergence, expansion outcomes, iterative == recursive tree building, dynamic trajectory lengths),
the Euclidean integrators' golden vectors, covariance formatting and the Gaussian-Euclidean metric (momentum EQUAL to
``mass_matrix_sqrt * random.normal``), Welford, dual averaging, ``run_inference_algorithm``, the diagnostics, the warm-up
schedule.  They pass -- which is the evidence that ``tests/golden/ref_shim_fixtures.json`` (the same code on the same
stand-in) records what the reference computes, up to fp32 rounding and the ``jax.random`` bit streams.

Runs in a subprocess (a module named ``jax`` must never be importable in this process), writes nothing into
``/root/reference``, and needs it: skipped on the GPU box."""
import importlib.util
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="/root/reference is not on this box")


def _runner():
    spec = importlib.util.spec_from_file_location("_refshim_runner", os.path.join(HERE, "refshim", "run_reference_tests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_unit_tests_pass_on_the_stand_in():
    runner = _runner()
    r = runner.run(runner.QUICK, timeout=1500)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 45, tail  # 48 tests (+ 82 parameter sets as subtests) at the time of writing
    assert " failed" not in tail and " error" not in tail.lower(), tail
    # nothing was left behind in the read-only reference tree
    assert not any("__pycache__" in d for d, _, _ in os.walk("/root/reference/tests"))


def test_a_module_named_jax_is_not_importable_here():
    """The stand-in lives under tests/refshim and is put on sys.path by the generator / runner subprocesses only: in this
    process (and in bench.py, smoke(), tools/rng_pin.py, which probe for a REAL jax) ``import jax`` must keep failing."""
    assert importlib.util.find_spec("jax") is None
