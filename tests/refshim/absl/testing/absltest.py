"""NOT absl.testing.absltest: ``TestCase`` and ``main`` only."""
import unittest

TestCase = unittest.TestCase


def main(*args, **kwargs):
    unittest.main()
