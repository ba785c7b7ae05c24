"""Self-verifying pin of the jax.random bit streams (VERDICT r2 "next" #7).

The build container and the GPU boxes of this project have no JAX, so the threefry / erf_inv
restatements (``blackjax_amd/csrc/bjx_device.h``, ``blackjax_amd.random``) are pinned on Random123
KATs and on the reference's own golden vectors only.  ``check(dev)`` closes that the first time it
runs on a box where ``import jax`` works: it regenerates the ``prng`` section of
``tests/golden/gen_jax_fixtures.py`` IN PROCESS and compares it with the product's host helpers and
with the device kernels (the momentum draw = ``normal(split(split(key, N)[c], 2)[0], (D,))``,
blackjax/util.py:90 through metrics.py:260-261; ``bjx_keys_randint`` = dynamic_hmc.py:69).

Returns one string for the bench / smoke JSON:
    "verified (jax X.Y.Z: ...)" | "mismatch: <what>" | "unavailable: <reason>"
Called by ``bench.py`` (outside every timed region) and ``__graft_entry__.smoke()``; never raises.
"""
from __future__ import annotations

import os


def _ulps(a, b):
    import numpy as np

    ai = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def _add_jax_site():
    """``BJX_JAX_SITE``: os.pathsep-separated site-packages directories of a JAX installed outside this
    interpreter's path (a driver-side venv); appended to sys.path, so nothing already importable is shadowed."""
    import sys

    for p in os.environ.get("BJX_JAX_SITE", "").split(os.pathsep):
        if p and os.path.isdir(p) and p not in sys.path:
            sys.path.append(p)


def check(dev=None) -> str:
    try:
        os.environ.setdefault("JAX_PLATFORMS", "cpu")
        _add_jax_site()
        import jax
        import jax.numpy as jnp
    except Exception as e:  # ModuleNotFoundError on every box seen so far
        return f"unavailable: import jax failed ({type(e).__name__}: {e})"
    try:
        import numpy as np
        import torch

        import blackjax_amd as bjx
        from blackjax_amd import _lib

        if not bool(jax.config.jax_threefry_partitionable):
            return "unavailable: this jax runs the legacy (non-partitionable) threefry layout"

        def words(k):
            return np.asarray(jax.random.key_data(k)).astype(np.uint32)

        bad = []
        n_checked = 0
        for seed in (0, 1, 42, 2024, (7 << 32) + 5):
            k = jax.random.key(seed)
            mine = bjx.random.key(seed)
            if not np.array_equal(words(k), mine):
                bad.append(f"key({seed})")
            for n in (2, 3, 5):
                if not np.array_equal(words(jax.random.split(k, n)), bjx.random.split(mine, n)):
                    bad.append(f"split(key({seed}), {n})")
            for d in (0, 1, 7, 1023):
                if not np.array_equal(words(jax.random.fold_in(k, d)), bjx.random.fold_in(mine, d)):
                    bad.append(f"fold_in(key({seed}), {d})")
            u = np.float32(jax.random.uniform(k, (), jnp.float32))
            if u != bjx.random.uniform(mine):
                bad.append(f"uniform(key({seed}))")
            n_checked += 9
            if dev is not None and torch.cuda.is_available():
                # device kernels: momentum draw with imm = 1 -> chain c gets normal(split(split(k, N)[c], 2)[0], (D,))
                N, D = 4, 1024
                p = torch.empty((N, D), dtype=torch.float32, device=dev)
                ke = torch.empty(N, dtype=torch.float32, device=dev)
                imm = torch.ones(D, dtype=torch.float32, device=dev)
                k0, k1 = bjx.random.key_words(mine)
                _lib.call("bjx_hmc_momentum_diag", _lib.current_stream(), k0, k1, 0, -1, N, D, imm.data_ptr(), 0,
                          p.data_ptr(), ke.data_ptr())
                got = p.cpu().numpy()
                ck = jax.random.split(k, N)
                ref = np.stack([np.asarray(jax.random.normal(jax.random.split(ck[c], 2)[0], (D,), jnp.float32))
                                for c in range(N)])
                # XLA's f32 log1p inside erf_inv is not correctly rounded, the kernels' is: <= 2 ulp,
                # and at most 1 % of the draws may differ at all (tests/test_jax_fixtures.py)
                ul = _ulps(got, ref)
                if ul.max() > 2 or np.mean(got != ref) > 0.01:
                    bad.append(f"device normal stream of key({seed}): max {int(ul.max())} ulp, "
                               f"{float(np.mean(got != ref)):.4f} of draws differ")
                n_checked += N * D
        if bad:
            return "mismatch: " + "; ".join(bad[:6]) + (f" (+{len(bad) - 6} more)" if len(bad) > 6 else "")
        return (f"verified (jax {jax.__version__}: key / split / fold_in / uniform words bit-exact, device "
                f"normal stream within 2 ulp; {n_checked} values)")
    except Exception as e:
        return f"unavailable: the comparison itself failed ({type(e).__name__}: {e})"


if __name__ == "__main__":
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch

    print(check(torch.device("cuda:0") if torch.cuda.is_available() else None))
