"""Error behaviour of the host-side mirror ("same names, argument meaning and error behaviour"):
the argument checks the reference makes BEFORE any traced code raise the same exception type here, with the same
message text.  The expected text is read from the reference's source (``ast``: the string fragments of the ``raise``
at the cited line), so the test follows the reference, not a transcription of it; it runs where ``/root/reference``
exists and needs no GPU (every check fires before a kernel would be launched)."""
import ast
import os
import re

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import dynamic_hmc, metrics, util

REF = "/root/reference/blackjax/"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not on this box")


def _raise_at(path, lineno):
    """-> (exception name, [literal message fragments]) of the ``raise`` statement that starts at ``path:lineno``."""
    with open(REF + path) as fh:
        tree = ast.parse(fh.read())
    node = next(n for n in ast.walk(tree) if isinstance(n, ast.Raise) and n.lineno == lineno)
    call = node.exc
    frags = []
    for part in ast.walk(call.args[0]):
        if isinstance(part, ast.Constant) and isinstance(part.value, str):
            frags.append(part.value)
    return call.func.id, [f for f in frags if len(f.strip()) > 3]


def _expect(path, lineno, fn):
    exc_name, frags = _raise_at(path, lineno)
    exc = {"ValueError": ValueError, "TypeError": TypeError, "NotImplementedError": NotImplementedError}[exc_name]
    with pytest.raises(exc) as ei:
        fn()
    msg = re.sub(r"\s+", " ", str(ei.value))
    for f in frags:  # every literal piece of the reference's message (the formatted values lie between them)
        assert re.sub(r"\s+", " ", f).strip() in msg, (f, msg)


def _fn(q):
    return -0.5 * (q * q).sum(-1)


def test_window_adaptation_argument_checks():
    """window_adaptation.py:401-423: shape of ``initial_inverse_mass_matrix`` against ``is_mass_matrix_diagonal`` and the
    sign of ``imm_shrinkage_to_previous``, checked when the warm-up is BUILT."""
    kw = dict(num_integration_steps=3)
    _expect("adaptation/window_adaptation.py", 405,
            lambda: bjx.window_adaptation(bjx.hmc, _fn, True, initial_inverse_mass_matrix=np.eye(3), **kw))
    _expect("adaptation/window_adaptation.py", 411,
            lambda: bjx.window_adaptation(bjx.hmc, _fn, False, initial_inverse_mass_matrix=np.ones(3), **kw))
    _expect("adaptation/window_adaptation.py", 411,
            lambda: bjx.window_adaptation(bjx.hmc, _fn, False, initial_inverse_mass_matrix=np.ones((3, 2)), **kw))
    _expect("adaptation/window_adaptation.py", 419,
            lambda: bjx.window_adaptation(bjx.hmc, _fn, imm_shrinkage_to_previous=-0.1, **kw))
    # the boundary value is accepted (>= 0.0)
    assert isinstance(bjx.window_adaptation(bjx.nuts, _fn, imm_shrinkage_to_previous=0.0), bjx.AdaptationAlgorithm)


def test_staged_adaptation_argument_checks():
    """staged_adaptation.py:666-676 (``n_chains``); ``metric="auto"`` and non-registry metrics are outside the path this
    engine replaces and say so (NotImplementedError) instead of the reference's budget check."""
    kw = dict(num_integration_steps=3)
    _expect("adaptation/staged_adaptation.py", 666, lambda: bjx.staged_adaptation(bjx.hmc, _fn, n_chains=0, **kw))
    _expect("adaptation/staged_adaptation.py", 668, lambda: bjx.staged_adaptation(bjx.hmc, _fn, n_chains=2, **kw))
    _expect("adaptation/staged_adaptation.py", 510, lambda: bjx.staged_adaptation(bjx.hmc, _fn, metric=3, **kw))
    with pytest.raises(NotImplementedError, match="auto"):
        bjx.staged_adaptation(bjx.hmc, _fn, metric="auto", max_grad_budget=50_000, **kw)
    with pytest.raises(NotImplementedError):
        bjx.staged_adaptation(bjx.hmc, _fn, metric="fisher_diag", **kw)
    for name in ("welford_diag", "welford_dense"):
        assert isinstance(bjx.staged_adaptation(bjx.nuts, _fn, name), bjx.AdaptationAlgorithm)


def test_run_inference_algorithm_argument_checks():
    """util.py:190-197: exactly one of ``initial_state`` / ``initial_position``."""
    alg = bjx.hmc(_fn, 0.1, torch.ones(3), 3)
    _expect("util.py", 191, lambda: util.run_inference_algorithm(bjx.random.key(0), alg, 3))
    _expect("util.py", 195, lambda: util.run_inference_algorithm(bjx.random.key(0), alg, 3, initial_state=1, initial_position=2))


def test_metric_dimension_check():
    """metrics.py:725-728: an inverse mass matrix that is neither 1-d nor 2-d."""
    _expect("mcmc/metrics.py", 725, lambda: metrics.default_metric(torch.tensor(1.0), 5, 2, torch.device("cpu")))


def test_halton_sequence_bit_width_check():
    """dynamic_hmc.py:210-213."""
    _expect("mcmc/dynamic_hmc.py", 211, lambda: dynamic_hmc.halton_sequence(np.arange(3, dtype=np.int32), max_bits=32))
    _expect("mcmc/dynamic_hmc.py", 211, lambda: dynamic_hmc.halton_sequence(torch.arange(3, dtype=torch.int32), max_bits=40))
