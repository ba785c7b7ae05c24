"""The C-ABI library loads without a GPU and exports every symbol include/bjx_hip.h declares."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bjx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bjx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from blackjax_amd import _lib

    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bjx_hip.h but not exported"
    # and every bound prototype is declared in the header
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in include/bjx_hip.h"
    assert lib.bjx_abi_version() == 1


def test_error_reporting_without_gpu():
    from blackjax_amd import _lib

    lib = _lib.load()
    rc = lib.bjx_leapfrog_diag(None, 4, 8, 3, 0.1, None, None, 0, None, None, None, None, None)
    assert rc != 0 and b"bjx_leapfrog_diag" in lib.bjx_last_error()


def test_host_key_split_matches_oracle():
    import blackjax_amd as bjx
    from oracle import prng

    k = bjx.random.key(2**33 + 5)
    assert np.array_equal(k, prng.key(2**33 + 5))
    assert np.array_equal(bjx.random.split(k, 9, offset=4), prng.split(k, 9, offset=4))
    assert np.array_equal(bjx.random.fold_in(k, 77), prng.fold_in(k, 77))
