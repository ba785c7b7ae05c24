"""NOT chex.  The few names the reference's own unit tests use (``chex.TestCase`` with ``self.variant``, ``all_variants`` /
``variants``, ``assert_trees_all_close`` ...), so that those tests can be collected and run on ``tests/refshim`` -- the
check that the stand-in for JAX is faithful enough for the reference's tests of the hot path to pass on it."""
from __future__ import annotations

import unittest

import numpy as np


class TestCase(unittest.TestCase):
    def variant(self, fn, **kwargs):  # with_jit / without_jit / with_device ...: one variant here, the function itself
        return fn


def _variants(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


all_variants = variants = _variants


def _leaves(tree):
    import jax

    return jax.tree.leaves(tree)


def assert_trees_all_close(*trees, rtol=1e-6, atol=0.0, **kwargs):
    first = _leaves(trees[0])
    for other in trees[1:]:
        o = _leaves(other)
        assert len(o) == len(first), "trees differ in structure"
        for a, b in zip(first, o):
            np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def assert_trees_all_equal(*trees, **kwargs):
    first = _leaves(trees[0])
    for other in trees[1:]:
        for a, b in zip(first, _leaves(other)):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def assert_tree_all_finite(tree):
    for a in _leaves(tree):
        assert np.all(np.isfinite(np.asarray(a)))


def assert_shape(x, shape):
    assert tuple(np.shape(x)) == tuple(shape), (np.shape(x), shape)


def assert_equal_shape(xs):
    shapes = {tuple(np.shape(x)) for x in xs}
    assert len(shapes) == 1, shapes


def assert_scalar(x):
    assert np.ndim(x) == 0


def clear_trace_counter():
    pass


def assert_max_traces(*args, **kwargs):
    return (lambda fn: fn) if not (len(args) == 1 and callable(args[0])) else args[0]
