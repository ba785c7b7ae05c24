import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch

    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Say ONCE why the JAX-generated fixture tests skip, instead of
    leaving it to -rs: no box of this project has a JAX wheel, so tests/golden/jax_fixtures.json has
    never been generated; the jax.random streams are pinned on the Random123 KATs and on six values
    printed in JAX's documentation (tests/test_oracle_prng.py, test_hmc_gpu.py::test_device_rng_*)."""
    skipped = [r for r in terminalreporter.stats.get("skipped", [])
               if "test_jax_fixtures" in getattr(r, "nodeid", "")]
    if skipped:
        reason = skipped[0].longrepr[2] if isinstance(skipped[0].longrepr, tuple) else str(skipped[0].longrepr)
        terminalreporter.write_line(
            f"a34: {len(skipped)} JAX-fixture test(s) skipped -- {reason} "
            "(generate with tests/golden/gen_jax_fixtures.py on a box where `import jax, blackjax` works; "
            "BJX_JAX_SITE=<site-packages> adds a JAX outside this interpreter's path)")
