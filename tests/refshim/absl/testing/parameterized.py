"""NOT absl.testing.parameterized: ``parameters`` / ``named_parameters`` / ``product`` as decorators that run the test body once
per parameter set inside ONE test method (``subTest`` per set) -- enough for the reference's tests."""
import functools
import itertools
import unittest

TestCase = unittest.TestCase


def _expand(sets):
    # absl: a single argument that is an iterable but neither a tuple nor a dict is the LIST of test cases
    if len(sets) == 1 and not isinstance(sets[0], (tuple, dict, str)) and hasattr(sets[0], "__iter__"):
        sets = tuple(sets[0])
    return sets


def _call(fn, self, p):
    if isinstance(p, dict):
        kw = {k: v for k, v in p.items() if k != "testcase_name"}
        return fn(self, **kw)
    if isinstance(p, (list, tuple)):
        return fn(self, *p)
    return fn(self, p)


def parameters(*sets):
    sets = _expand(sets)

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(self):
            for p in sets:
                with self.subTest(params=repr(p)[:80]):
                    _call(fn, self, p)

        return wrapped

    return deco


def named_parameters(*sets):
    sets = _expand(sets)

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(self):
            for p in sets:
                if isinstance(p, (list, tuple)):  # ("name", arg, ...)
                    name, args = p[0], p[1:]
                    with self.subTest(name=name):
                        fn(self, *args)
                else:
                    with self.subTest(name=p.get("testcase_name")):
                        _call(fn, self, p)

        return wrapped

    return deco


def product(*args, **kwargs):
    keys = list(kwargs)
    combos = [dict(zip(keys, vals)) for vals in itertools.product(*[kwargs[k] for k in keys])]
    return parameters(*combos)
