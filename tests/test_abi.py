"""The C-ABI library loads without a GPU and exports every symbol include/bjx_hip.h declares."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bjx_hip.h")).read()
    text += open(os.path.join(ROOT, "include", "bjx_nuts.h")).read()
    text += open(os.path.join(ROOT, "include", "bjx_pool.h")).read()
    text += open(os.path.join(ROOT, "include", "bjx_ghmc.h")).read()
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
    for s in list(_lib.SIGNATURES) + list(_lib.INT64_FUNCTIONS):
        assert s in syms, f"{s} bound in _lib.py but not declared in include/*.h"
    assert lib.bjx_pool_workspace_bytes(0, 8) == 0
    assert lib.bjx_pool_workspace_bytes(65536, 1024) == 512 * 4 * 1024 * 8  # 512 slabs x K=4 x D doubles
    assert lib.bjx_abi_version() == 7  # 7 (round 6): + bjx_log1p_device_check


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


def test_nuts_descriptor_and_slots_match_header():
    """The ctypes mirror of bjx_nuts_t and the slot tables agree with include/bjx_nuts.h."""
    from blackjax_amd import _lib

    text = open(os.path.join(ROOT, "include", "bjx_nuts.h")).read()
    text_nc = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    enums = dict((k, int(v)) for k, v in re.findall(r"(BJX_NUTS_[A-Z_0-9]+)\s*=\s*(\d+)", text_nc))
    for name, i in _lib.NUTS_F.items():
        assert enums["BJX_NUTS_F_" + name] == i
    for name, i in _lib.NUTS_I.items():
        assert enums["BJX_NUTS_I_" + name] == i
    assert enums["BJX_NUTS_NF"] == _lib.NUTS_NF == len(_lib.NUTS_F)
    assert enums["BJX_NUTS_NI"] == _lib.NUTS_NI == len(_lib.NUTS_I)
    # field order of the struct
    body = re.search(r"typedef struct \{(.*?)\} bjx_nuts_t;", text_nc, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int64_t|int32_t|uint32_t|float)\s*\*?", "", decl)
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    py = [f[0].rstrip("_") for f in _lib.NutsDesc._fields_]
    assert names == py, (names, py)
    assert ctypes.sizeof(_lib.NutsDesc) == 8 * 2 + 4 * 2 + 8 * 3 + 4 * 2 + 4 * 2 + 8 * 2 + 8 * 19 + 8 * 6 + 4 * 2 + 8
    # the free-running run descriptor
    body = re.search(r"typedef struct \{([^}]*?)\} bjx_nuts_async_t;", text_nc, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int64_t|int32_t|uint32_t|uint8_t|float)\s*\*?", "", decl)
        names += [re.sub(r"\[.*\]$", "", n.strip().lstrip("*")) for n in decl.split(",")]
    py = [f[0] for f in _lib.NutsAsync._fields_]
    assert names == py, (names, py)
    max_mid = int(re.search(r"#define BJX_NUTS_MAX_MID (\d+)", text_nc).group(1))
    assert max_mid == _lib.NUTS_MAX_MID == len(_lib.NutsAsync().int_mid_kick) == len(_lib.NutsAsync().int_mid_drift)
    assert ctypes.sizeof(_lib.NutsAsync) == (8 + 4 * 2 + 8 * 7 + 8 * 2 + 8 * 8 + 8 + 4 * 2 + 8 * 9 + 8 * 4 + 4 * 2 + 8
                                             + 8 * 2 + 4 * 2 + 8 + 4 * 2 + 4 * 2 * max_mid + 8 * 8)
    for name, i in _lib.NUTS_AT.items():
        assert enums["BJX_NUTS_AT_" + name] == i
    assert enums["BJX_NUTS_ADAPT_COLS"] == _lib.NUTS_ADAPT_COLS


def test_spec_tail_descriptor_matches_header():
    """ctypes mirror of bjx_nuts_spec_t (the two-stream speculative tail, ABI 6)."""
    from blackjax_amd import _lib

    text = open(os.path.join(ROOT, "include", "bjx_nuts.h")).read()
    text_nc = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct \{([^}]*?)\} bjx_nuts_spec_t;", text_nc, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int64_t|int32_t|uint32_t|float)\s*\*?", "", decl)
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in _lib.NutsSpec._fields_], names
    assert ctypes.sizeof(_lib.NutsSpec) == 8 * 3 + 4 * 2 + 8 * 17
    assert int(re.search(r"#define BJX_NUTS_SPEC_IW (\d+)", text_nc).group(1)) == _lib.NUTS_SPEC_IW
    assert int(re.search(r"#define BJX_NUTS_SPEC_TAG (\d+)", text_nc).group(1)) == _lib.NUTS_SPEC_TAG
    lib = _lib.load()  # argument checking happens before any device work
    assert lib.bjx_nuts_spec_book(None, None, None, None, 0, 0) != 0 and b"bjx_nuts_spec_book" in lib.bjx_last_error()
