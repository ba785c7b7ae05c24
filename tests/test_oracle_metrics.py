"""The reference's own metric tests (tests/mcmc/test_metrics.py:21-205: ``_format_covariance`` and
``gaussian_euclidean``) restated for the oracle AND for the product's host-side factorisation, on the
CPU: the wiring ``p = mass_matrix_sqrt (.) normal(key, (D,))`` with ``mass_matrix_sqrt = 1 / sqrt(imm)`` (diagonal) or
``L^{-T}``, ``L = cholesky(imm, lower)`` (dense), the kinetic energy ``0.5 <imm p, p>``, the error for a matrix that is
neither 1-d nor 2-d.  The product's ``blackjax_amd.metrics.default_metric`` builds its factor with torch on whatever
device the matrix lives on, so the factor the GEMM kernels are handed is checked here without a GPU."""
import numpy as np
import pytest
import torch

from blackjax_amd import metrics as bmetrics
from oracle import hmc as ohmc
from oracle import prng

f32, f64 = np.float32, np.float64
IMM2 = np.asarray([[2 / 3, 0.5], [0.5, 3 / 4]], dtype=f32)  # test_metrics.py:62,167


@pytest.mark.parametrize("shape", [(), (1, 2, 3)])
def test_invalid_number_of_dimensions(shape):
    """test_metrics.py:27-40,108-119: 0-d and 3-d (non-square trailing) inputs."""
    with pytest.raises(ValueError, match="The mass matrix has the wrong number of dimensions"):
        ohmc.default_metric(np.ones(shape, f32))
    with pytest.raises(ValueError, match="The mass matrix has the wrong number of dimensions"):
        bmetrics.default_metric(torch.ones(shape), 5, 3, torch.device("cpu"))


def test_gaussian_euclidean_dim_1():
    """test_metrics.py:121-158: imm = [1/4] -> momentum = 2 * normal(key), exactly; K = 0.5 * (imm p) p, exactly."""
    key = prng.key(0)
    imm = np.asarray([1 / 4], dtype=f32)
    m = ohmc.default_metric(imm)
    assert not m.is_dense and m.mass_matrix_sqrt[0] == f32(2.0)  # 2 is the square root inverse of 1/4
    p = ohmc.sample_momentum(m, key[None], 1)
    assert p[0, 0] == f32(2.0) * prng.normal(key, ())
    k = ohmc.kinetic_energy(m, p)
    assert k[0] == f32(0.5) * (imm[0] * p[0, 0]) * p[0, 0]
    # scale(): p / sqrt(imm) and p * sqrt(imm) (test_metrics.py:147-158) are the factor and its inverse
    np.testing.assert_allclose(p[0] * m.mass_matrix_sqrt, p[0] / np.sqrt(imm), rtol=1e-6)
    # the product classifies the same input as a shared diagonal and keeps the values
    pm = bmetrics.default_metric(torch.as_tensor(imm), 7, 1, torch.device("cpu"))
    assert pm.kind == "diag" and pm.imm_stride == 0 and torch.equal(pm.imm, torch.as_tensor(imm))


def test_gaussian_euclidean_dim_2():
    """test_metrics.py:160-205: imm 2 x 2 -> momentum = inv(cholesky(imm, upper)) @ normal(key, (2,))."""
    key = prng.key(0)
    m = ohmc.default_metric(IMM2)
    assert m.is_dense
    p = ohmc.sample_momentum(m, key[None], 2)[0]
    U = np.linalg.cholesky(IMM2.astype(f64)).T          # linalg.cholesky(imm, lower=False)
    L_inv = np.linalg.inv(U)
    np.testing.assert_allclose(p, L_inv @ prng.normal(key, (2,)).astype(f64), rtol=1e-6)
    k = ohmc.kinetic_energy(m, p[None])[0]
    np.testing.assert_allclose(k, 0.5 * (IMM2.astype(f64) @ p.astype(f64)) @ p.astype(f64), rtol=1e-6)
    # scale(inv=False) = L_inv @ p (test_metrics.py:197-205): the factor itself
    np.testing.assert_allclose(m.mass_matrix_sqrt, L_inv, rtol=1e-6)


def test_format_covariance_dim_2_identities():
    """test_metrics.py:58-100 (is_inv=True): sqrt sqrt^T = inv(imm), inv_sqrt inv_sqrt^T = imm -- oracle and product."""
    m = ohmc.default_metric(IMM2)
    S = m.mass_matrix_sqrt.astype(f64)
    np.testing.assert_allclose(S @ S.T, np.linalg.inv(IMM2.astype(f64)), rtol=1e-6)
    pm = bmetrics.default_metric(torch.as_tensor(IMM2), 4, 2, torch.device("cpu"))
    assert pm.kind == "dense"
    Sp = pm.mass_sqrt_t.numpy().astype(f64).T             # the product stores the transpose (row-major GEMM operand)
    np.testing.assert_allclose(Sp @ Sp.T, np.linalg.inv(IMM2.astype(f64)), rtol=1e-6)
    inv_sqrt = np.linalg.inv(Sp).T                        # = L (lower): the reference's inv_mass_matrix_sqrt
    np.testing.assert_allclose(inv_sqrt @ inv_sqrt.T, IMM2, rtol=1e-6)
    assert np.all(np.triu(inv_sqrt, 1) == 0)
    assert torch.equal(pm.imm_t, torch.as_tensor(IMM2).t())


@pytest.mark.parametrize("D,rho", [(2, 0.5), (64, 0.9), (512, 0.9)])
def test_product_factor_equals_oracle_factor(D, rho):
    """SURVEY row a7 on the CPU: the factor the product hands its GEMM kernels (fp64 Cholesky + triangular inverse, rounded
    once) against the oracle's own -- two LAPACK builds (torch / NumPy) may round a handful of entries differently, hence
    one fp32 ulp; the AR(1) matrix at D = 512, rho = 0.9 is BASELINE.json's C5 target."""
    idx = np.arange(D)
    cov = (rho ** np.abs(idx[:, None] - idx[None, :])).astype(f32)
    mo = ohmc.default_metric(cov)
    pm = bmetrics.default_metric(torch.as_tensor(cov), 16, D, torch.device("cpu"))
    got, ref = pm.mass_sqrt_t.numpy().T, mo.mass_matrix_sqrt
    assert got.dtype == f32 and ref.dtype == f32
    np.testing.assert_allclose(got, ref, rtol=2.5e-7, atol=1e-9)
    assert np.mean(got == ref) > 0.9  # (measured: 94-100 % of the entries are the same float, the rest one ulp apart)
    # upper-triangular (S = L^{-T}); S S^T = imm^{-1}, i.e. with L = S^{-T}: L L^T = imm
    assert np.all(np.tril(got, -1) == 0)
    L = np.linalg.inv(got.astype(f64)).T
    np.testing.assert_allclose(L @ L.T, cov, rtol=0, atol=5e-6)


def test_per_chain_dense_factor():
    """A (N, D, D) stack (what a vmapped dense warm-up returns): one factor per chain, each equal to the 2-d result."""
    rng = np.random.default_rng(0)
    A = rng.standard_normal((3, 6, 6))
    cov = (A @ np.swapaxes(A, 1, 2) + 6 * np.eye(6)).astype(f32)
    pm = bmetrics.default_metric(torch.as_tensor(cov), 3, 6, torch.device("cpu"))
    mo = ohmc.default_metric(cov)
    assert pm.kind == "dense_pc"
    for c in range(3):
        one = bmetrics.default_metric(torch.as_tensor(cov[c].copy()), 3, 6, torch.device("cpu"))
        assert torch.equal(pm.mass_sqrt_t[c], one.mass_sqrt_t)
        np.testing.assert_allclose(pm.mass_sqrt_t[c].numpy().T, mo.mass_matrix_sqrt[c], rtol=2.5e-7, atol=1e-9)
