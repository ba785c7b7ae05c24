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
