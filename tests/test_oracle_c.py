"""The C/OpenMP port of the oracle (bench cpu_baseline) is bit-identical to the NumPy oracle."""
import numpy as np

from oracle import cport, hmc as ohmc, prng, targets


def test_c_port_bit_identical():
    N, D, L = 48, 200, 7
    rng = np.random.default_rng(3)
    imm = rng.uniform(0.2, 3.0, D).astype(np.float32)
    iv = rng.uniform(0.2, 3.0, D).astype(np.float32)
    fn = targets.diag_gaussian(iv)
    st = ohmc.init(prng.normal(prng.key(1), (N, D)), fn)
    q, lp, g = st.position.copy(), st.logdensity.copy(), st.logdensity_grad.copy()
    n_rej = 0
    for k in prng.split(prng.key(0), 6):
        st, info = ohmc.kernel(k, st, fn, np.float32(0.45), imm, L, chain_offset=17)
        acc, ia, idv = cport.hmc_diag_gaussian_step(k, q, lp, g, 0.45, imm, iv, L, chain_offset=17)
        assert np.array_equal(ia, info.is_accepted)
        assert np.array_equal(idv, info.is_divergent)
        assert np.array_equal(acc, info.acceptance_rate)
        assert np.array_equal(q, st.position) and np.array_equal(g, st.logdensity_grad)
        assert np.array_equal(lp, st.logdensity)
        n_rej += int((~ia).sum())
    assert n_rej > 0


def test_c_port_per_chain_variant_bit_identical():
    """Explicit chain keys (an arbitrary subset of global chain indices), per-chain step sizes and
    per-chain inverse mass diagonals: the C port equals oracle/hmc.py::kernel with
    chain_keys_override bit for bit."""
    N, D, L = 23, 77, 5
    rng = np.random.default_rng(4)
    imm = rng.uniform(0.2, 3.0, (N, D)).astype(np.float32)
    eps = rng.uniform(0.05, 0.6, N).astype(np.float32)
    iv = rng.uniform(0.2, 3.0, D).astype(np.float32)
    fn = targets.diag_gaussian(iv)
    idx = rng.choice(1 << 40, N, replace=False)
    st = ohmc.init(prng.normal(prng.key(2), (N, D)), fn)
    q, lp, g = st.position.copy(), st.logdensity.copy(), st.logdensity_grad.copy()
    for k in prng.split(prng.key(5), 4):
        keys = prng.split_at(k, idx)
        assert np.array_equal(keys[3], prng.split(k, 1, offset=int(idx[3]))[0])
        st, info = ohmc.kernel(None, st, fn, eps, imm, L, chain_keys_override=keys, per_chain_diag=True)
        acc, ia, idv = cport.hmc_diag_gaussian_step_pc(keys, q, lp, g, eps, imm, iv, L)
        assert np.array_equal(ia, info.is_accepted) and np.array_equal(acc, info.acceptance_rate)
        assert np.array_equal(q, st.position) and np.array_equal(g, st.logdensity_grad)
        assert np.array_equal(lp, st.logdensity)
    # shared (D,) diagonal through the same entry point
    st2 = ohmc.init(prng.normal(prng.key(3), (N, D)), fn)
    q, lp, g = st2.position.copy(), st2.logdensity.copy(), st2.logdensity_grad.copy()
    keys = prng.split_at(prng.key(9), idx)
    st2, info = ohmc.kernel(None, st2, fn, eps, imm[0], L, chain_keys_override=keys)
    acc, ia, _ = cport.hmc_diag_gaussian_step_pc(keys, q, lp, g, eps, imm[0], iv, L)
    assert np.array_equal(ia, info.is_accepted) and np.array_equal(q, st2.position)


def test_f32chain_gemm_c_equals_numpy_and_differs_from_fp64_rounding():
    """The dense-metric "f32 chain" mode: the C fmaf-chain GEMM equals the NumPy statement
    (one correctly rounded fma32 per k, in fp.mfma_k_order) bit for bit, for a strided B view, and
    is a genuinely different rounding from the fp64-accumulated product."""
    from oracle import fp

    rng = np.random.default_rng(0)
    for M, K in [(5, 6), (9, 40), (4, 130)]:
        a = rng.standard_normal((M, K)).astype(np.float32)
        mat = rng.standard_normal((K, K)).astype(np.float32)
        order = fp.mfma_k_order(K)
        assert sorted(order.tolist()) == list(range(K))
        ref = fp.gemm_f32chain(a, mat.T, order)
        got = cport.gemm_f32chain(a, mat.T, order)
        assert np.array_equal(got, ref)
        exact = (a.astype(np.float64) @ mat.astype(np.float64).T)
        assert np.abs(got - exact).max() < 1e-4
    assert fp.mfma_k_order(20).tolist() == [0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15, 16, 17, 18, 19]
    assert np.any(got != exact.astype(np.float32))  # K = 130: the fp32 chain rounds differently somewhere


def test_dense_f32chain_mode_of_the_kernel():
    """oracle/hmc.py::kernel with a dense metric in f32-chain mode stays within fp32 round-off of the
    fp64-accumulated mode (same accept decisions on this small case) and is deterministic."""
    N, D, L = 12, 24, 6
    cov = targets.ar1_covariance(0.9, D)
    fn = targets.ar1_gaussian(0.9, D)
    st = ohmc.init(prng.normal(prng.key(1), (N, D)), fn)
    m64 = ohmc.default_metric(cov, n_chains=N)
    m32 = ohmc.default_metric(cov, n_chains=N, dense_accum="f32chain")
    k = prng.key(3)
    s64, i64 = ohmc.kernel(k, st, fn, np.float32(0.4), cov, L, metric=m64)
    s32, i32 = ohmc.kernel(k, st, fn, np.float32(0.4), cov, L, metric=m32)
    s32b, _ = ohmc.kernel(k, st, fn, np.float32(0.4), cov, L, metric=m32)
    assert np.array_equal(s32.position, s32b.position)
    np.testing.assert_allclose(i32.proposal.position, i64.proposal.position, rtol=1e-4, atol=1e-4)
    assert np.array_equal(i32.is_accepted, i64.is_accepted)
