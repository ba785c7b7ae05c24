"""The golden vectors of tests/golden/reference_kats.json are TRANSCRIBED from the reference's own tests (the
reference cannot run here: no JAX).  Where ``/root/reference`` exists this checks the transcription against the source:
every number of a golden entry occurs as a numeric literal in the test file its ``source`` field cites (sign aside:
``-1.5`` is ``USub(1.5)`` in the syntax tree), inside the cited line range."""
import ast
import json
import os
import re

import pytest

REF_TESTS = "/root/reference/"
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS + "tests"), reason="/root/reference is not on this box")

with open(os.path.join(HERE, "golden", "reference_kats.json")) as _fh:
    KATS = json.load(_fh)


def _literals(source, whole_file=False):
    """Numeric literals (absolute values) of the cited file, restricted to the cited line ranges."""
    m = re.match(r"(tests/[\w/]+\.py):([\d,\-]+)", source)
    path, ranges = m.group(1), [tuple(int(x) for x in r.split("-")) for r in m.group(2).split(",")]
    with open(REF_TESTS + path) as fh:
        tree = ast.parse(fh.read())
    out = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)) and not isinstance(n.value, bool):
            if whole_file or any(lo <= n.lineno <= (hi[0] if hi else lo) for lo, *hi in ranges):
                out.add(abs(float(n.value)))
    return out


def _numbers(obj):
    if isinstance(obj, bool):
        return
    if isinstance(obj, (int, float)):
        yield abs(float(obj))
    elif isinstance(obj, (list, tuple)):
        for x in obj:
            yield from _numbers(x)
    elif isinstance(obj, dict):
        for x in obj.values():
            yield from _numbers(x)


def test_velocity_verlet_golden_vectors_are_the_reference_literals():
    e = KATS["velocity_verlet_mvnormal"]
    lits = _literals(e["source"])
    for field in ("q_init", "p_init", "cov", "q_final", "p_final", "step_size", "num_steps"):
        missing = [v for v in _numbers(e[field]) if v not in lits]
        assert not missing, (field, missing)
    assert len(list(_numbers(e["cov"]))) == 36 and len(e["q_final"]) == len(e["p_final"]) == 6


def test_uturn_truth_table_is_the_reference_literals():
    e = KATS["iterative_uturn"]
    lits = _literals(e["source"])
    for field in ("momentum", "momentum_sum", "momentum_ckpts", "momentum_sum_ckpts", "cases"):
        missing = [v for v in _numbers(e[field]) if v not in lits]
        assert not missing, (field, missing)


def test_expansion_cases_are_the_reference_literals():
    e = KATS["dynamic_expansion"]
    lits = _literals(e["source"])
    for case in e["cases"]:
        assert abs(float(case[0])) in lits and float(case[3]) in lits, case
    assert float(e["max_doublings"]) in lits
    # (``divergence_threshold`` is a module-level constant of that test file)
    assert float(e["divergence_threshold"]) in _literals(e["source"], whole_file=True)


def test_schedule_lengths_are_the_reference_literals():
    e = KATS["build_schedule"]
    lits = _literals(e["source"])
    for num_steps in ("19", "100", "200"):
        assert float(num_steps) in lits
        assert sum(c for _, _, c in e[num_steps]) == int(num_steps)
    # the window boundaries the reference's test spells out (its expected schedule is built from these counts)
    counted = {float(c) for k in ("100", "200") for _, _, c in e[k]}
    assert counted & lits, (counted, sorted(lits))


def test_dual_averaging_and_welford_parameters_are_the_reference_literals():
    e = KATS["dual_averaging"]
    lits = _literals(e["source"])
    for field in ("gamma", "x_init", "num_updates", "expected_final", "delta"):
        assert abs(float(e[field])) in lits, (field, e[field], sorted(lits))
    w = KATS["welford"]
    lits = _literals(w["source"])
    for field in ("numpy_seed", "num_samples", "rtol"):
        assert abs(float(w[field])) in lits, (field, w[field], sorted(lits))
