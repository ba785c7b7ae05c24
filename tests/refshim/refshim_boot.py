"""pytest plugin (``-p refshim_boot``) for running the reference's own unit tests on the stand-in: no bytecode written into
the read-only reference tree, and the ``blackjax._version`` module setuptools_scm would have generated."""
import sys
import types

sys.dont_write_bytecode = True
_v = types.ModuleType("blackjax._version")
_v.__version__ = "reference-source-tree"
sys.modules.setdefault("blackjax._version", _v)
