"""CPU oracle: a NumPy restatement of the BlackJAX HMC/NUTS hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``blackjax_amd`` never imports, links or executes anything from here; the
product path fails loudly when the HIP library is missing.

What it restates (each function cites the reference file:line it follows):

* ``prng``         JAX's threefry2x32 key / split / fold_in / bits / uniform /
                   normal / bernoulli (jax==0.10.0, ``jax_threefry_partitionable``
                   default) -- third-party arithmetic that is NOT under
                   ``/root/reference`` (pinned in ``uv.lock``).
* ``hmc``          ``blackjax/mcmc/{hmc,integrators,metrics,proposal,trajectory}.py``
* ``nuts``         ``blackjax/mcmc/{nuts,trajectory,termination,proposal}.py``
* ``adaptation``   ``blackjax/optimizers/dual_averaging.py``,
                   ``blackjax/adaptation/{step_size,mass_matrix,staged_adaptation,
                   window_adaptation}.py``
* ``diagnostics``  ``blackjax/diagnostics.py`` (``effective_sample_size``)

Pinning status
--------------
* Deterministic arithmetic is pinned against the reference's own golden
  vectors / KATs (``tests/golden/*.json``: velocity-verlet end points from
  ``tests/mcmc/test_integrators.py``, the U-turn truth table from
  ``tests/mcmc/test_uturn.py``, the warmup schedules from
  ``tests/adaptation/test_adaptation.py``, the dual-averaging fixed point from
  ``tests/optimizers/test_optimizers.py``, Welford covariance recovery from
  ``tests/adaptation/test_mass_matrix.py``).
* The threefry block function is pinned against the Random123 known-answer
  vectors.
* The *bit stream* of ``jax.random.normal`` / ``uniform`` built on top of it is
  **parity unpinned**: the reference's tests hold no literal expected outputs
  of ``jax.random.*`` and JAX itself cannot be imported in this container
  (no wheel, no network, Python 3.10).  The layout follows jax 0.10.0's
  ``jax/_src/prng.py`` / ``random.py`` as documented in ``prng.py``.

Floating-point conventions (shared with the HIP kernels):
fp32 storage; every ``a + s*b`` update is one fused multiply-add (what XLA:CPU
emits for the reference on an FMA machine); every reduction (kinetic energy,
U-turn dot products, log-density of the built-in targets) is accumulated in
fp64 and rounded once to fp32; scalar transcendentals (exp, log, log1p,
logaddexp, expit) are evaluated in fp64 and rounded once to fp32.  These make
the result independent of reduction order and libm flavour, which is what lets
the GPU/CPU comparison be bit-exact on accept/reject decisions.
"""
