"""THE stated tolerance of the dense-metric path (VERDICT r2 "next" #4; NOTEBOOK.md section 3 item 6).

The reference applies a dense metric with ``jnp.dot(..., precision="highest")`` (util.py:23-61), an
fp32 dot whose summation ORDER is unspecified.  The engine's shared-matrix path is an fp32 fmaf chain in
one stated k order (bit-exact against the oracle's "f32chain" mode, tests/test_dense_gpu.py,
test_full_shape_gpu.py::test_c5_*); the oracle's "f64" mode is the order-independent reading of the
same dot.  This CPU test runs BOTH oracle modes at configs[4]'s shape (512-dim AR(1) Gaussian, L = 20,
eps = 0.5; 512 of the 16 384 chains, every 32nd, with the keys those chains have in the full
ensemble) for ten consecutive transitions WITHOUT re-synchronisation and asserts the bound that
DESIGN.md quotes:
    accept/reject decisions: identical (0 flips in 5 120 chain-transitions),
    positions after 10 transitions: max |dq| <= 2e-4 (measured 2.9e-5); acceptance probabilities within 5e-4
    (measured 1.8e-4: the energies are O(D) = a few hundred, one fp32 ulp of an energy is 3e-5).
It also factors ``cov`` twice independently (NumPy's LAPACK and torch's) to show how little of the
result rides on which fp64 Cholesky produced the fp32 factor -- the GPU tests hand the oracle the
engine's factor; tests/test_full_shape_gpu.py::test_c5_engine_vs_independent_factor counts the
same thing for the HIP path against a factor the engine never saw.
"""
import numpy as np

from oracle import hmc as ohmc
from oracle import prng
from oracle import targets as otargets

f32 = np.float32
N_FULL, D, L, EPS, RHO, T = 16384, 512, 20, 0.5, 0.9, 10


def _run(metric, cov, fn, q0, idx, keys):
    st = ohmc.init(q0, fn)
    acc, rate = [], []
    for k in keys:
        st, info = ohmc.kernel(None, st, fn, f32(EPS), cov, L, metric=metric,
                               chain_keys_override=prng.split_at(k, idx))
        acc.append(info.is_accepted.copy())
        rate.append(info.acceptance_rate.copy())
    return st.position, np.stack(acc), np.stack(rate)


def test_dense_f32chain_vs_f64_at_c5_shape_ten_transitions():
    cov = otargets.ar1_covariance(RHO, D)
    fn = otargets.ar1_gaussian(RHO, D)
    idx = np.arange(0, N_FULL, 32)
    q0 = np.random.default_rng(12).standard_normal((N_FULL, D), dtype=f32)[idx]
    keys = prng.split(prng.key(21), T)
    m32 = ohmc.default_metric(cov, dense_accum="f32chain")
    m64 = ohmc.default_metric(cov, dense_accum="f64")
    q32, a32, r32 = _run(m32, cov, fn, q0, idx, keys)
    q64, a64, r64 = _run(m64, cov, fn, q0, idx, keys)
    flips = int((a32 != a64).sum())
    dq = float(np.abs(q32 - q64).max())
    dr = float(np.abs(r32 - r64).max())
    print(f"f32chain vs f64, {len(idx)} chains x {T} transitions: accept flips {flips}, max|dq| {dq:.3e}, "
          f"max|d acceptance_rate| {dr:.3e}, rejections {int((~a64).sum())}")
    assert (~a64).sum() > 0  # the comparison saw both outcomes
    assert flips == 0
    assert dq <= 2e-4 and dr <= 5e-4

    # an independently computed factor (torch's LAPACK instead of NumPy's): a handful of the D^2 fp32
    # entries may round differently; what that does to ten transitions of the f32-chain mode
    import torch

    Lt = torch.linalg.cholesky(torch.as_tensor(cov, dtype=torch.float64))
    Linv = torch.linalg.solve_triangular(Lt, torch.eye(D, dtype=torch.float64), upper=False)
    factor_t = Linv.T.contiguous().numpy().astype(f32)  # L^{-T}
    n_diff = int((factor_t != m32.mass_matrix_sqrt).sum())
    # entries that are structurally zero come out as 1e-17-size noise in both builds: compare on the
    # scale of the matrix, not in ulps of each entry
    err = np.abs(factor_t.astype(np.float64) - m32.mass_matrix_sqrt.astype(np.float64)).max()
    scale = float(np.abs(m32.mass_matrix_sqrt).max())
    mt = ohmc.default_metric(cov, dense_accum="f32chain", mass_matrix_sqrt=factor_t)
    qt, at, rt = _run(mt, cov, fn, q0, idx, keys)
    flips_f = int((at != a32).sum())
    dq_f = float(np.abs(qt - q32).max())
    print(f"NumPy vs torch factor: {n_diff} of {D * D} entries differ (max |diff| {err:.2e} on a scale of "
          f"{scale:.2f}); accept flips {flips_f}, max|dq| {dq_f:.3e}")
    assert err <= 2.0 ** -23 * scale
    assert flips_f == 0 and dq_f <= 2e-4
