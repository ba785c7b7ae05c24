"""GPU parity of the dense-metric path (fp32 MFMA GEMMs) vs the oracle.

A shared dense matrix is applied on v_mfma_f32_32x32x2_f32: bit for bit an fp32 fmaf chain per
output element in a fixed k order (0,8,1,9,...,7,15 inside every K-tile of 16, the same in both GEMM
kernels).  The oracle's ``dense_accum="f32chain"`` mode restates exactly that arithmetic
(oracle/fp.py::mfma_k_order, oracle/cport.py::gemm_f32chain), so this path is compared BIT FOR BIT:
accept/reject decisions, momenta and positions, over consecutive transitions without re-syncing
the two sides.  Both sides are handed the same fp32 Cholesky factor (the engine's, checked to be
within 1 ulp of NumPy's: two fp64 LAPACK builds may round a few of its D^2 entries differently)."""
import json
import os

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import _lib
from oracle import cport, fp as ofp, hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


@pytest.mark.parametrize("N,D", [(300, 512), (7, 6), (129, 130), (1, 4), (256, 128), (33, 17), (128, 256)])
def test_dense_matmul_is_the_k_ordered_fp32_fma_chain(dev, N, D):
    """C = A @ B on the MFMA equals, bit for bit, the fmaf chain in oracle/fp.py::mfma_k_order
    (general kernel: ragged and complete tiles)."""
    g = torch.Generator(device=dev)
    g.manual_seed(N * 1000 + D)
    A = torch.randn(N, D, device=dev, generator=g)
    B = torch.randn(D, D, device=dev, generator=g)
    C = torch.full((N, D), float("nan"), device=dev)
    _lib.call("bjx_dense_matmul", _lib.current_stream(), N, D, A.data_ptr(), B.data_ptr(), C.data_ptr())
    ref = cport.gemm_f32chain(t2n(A), t2n(B), ofp.mfma_k_order(D))
    assert np.array_equal(t2n(C), ref)
    exact = (A.double() @ B.double())
    scale = (A.double().abs() @ B.double().abs())
    assert float(((C.double() - exact).abs() / scale).max()) < 1e-6


def _shared_factor(dev, cov, N, D):
    """The engine's fp32 factor L^{-1} of ``cov`` (as the oracle's mass_matrix_sqrt = L^{-T}),
    checked against NumPy's own factorisation to 1 ulp."""
    m = bjx.metrics.default_metric(dev_t(cov, dev), N, D, dev)
    mass_sqrt = np.ascontiguousarray(t2n(m.mass_sqrt_t).T)
    ref = ohmc.default_metric(cov, n_chains=N).mass_matrix_sqrt
    np.testing.assert_allclose(mass_sqrt, ref, rtol=2.5e-7, atol=1e-9)
    return ohmc.default_metric(cov, n_chains=N, dense_accum="f32chain", mass_matrix_sqrt=mass_sqrt)


def test_reference_golden_velocity_verlet_through_dense_kernels(dev):
    """tests/mcmc/test_integrators.py:74-103,136-145 of the reference: dense 6-d Gaussian, 16 steps."""
    k = KATS["velocity_verlet_mvnormal"]
    cov = torch.tensor(k["cov"], device=dev)
    P = torch.linalg.inv(cov.double())
    q = torch.tensor([k["q_init"]], device=dev)
    p = torch.tensor([k["p_init"]], device=dev)
    grad = lambda x: (-(x.double() @ P.T)).float()
    g = grad(q)
    s = _lib.current_stream()
    eps = k["step_size"]
    for i in range(k["num_steps"]):
        p_new = torch.empty_like(p)
        _lib.call("bjx_leapfrog_dense", s, 1, 6, 1 if i == 0 else 2, eps, None, cov.data_ptr(),
                  q.data_ptr(), p.data_ptr(), g.data_ptr(), q.data_ptr(), p_new.data_ptr())
        p = p_new
        g = grad(q)
    p = p + (eps * 0.5) * g  # closing half kick
    np.testing.assert_allclose(t2n(q)[0], k["q_final"], atol=k["atol"])
    np.testing.assert_allclose(t2n(p)[0], k["p_final"], atol=k["atol"])


@pytest.mark.parametrize("N,D,L", [(100, 64, 8), (37, 30, 5), (256, 128, 4), (128, 256, 3)])
def test_dense_hmc_vs_oracle_bit_exact(dev, N, D, L):
    """Scaled-down configs[4]: AR(1) correlated Gaussian, dense imm = Sigma, 10 CONSECUTIVE
    transitions with no re-sync.  N and D multiples of 128 take the k-contiguous ("TN") GEMM kernel,
    the others the general one; both sum in the same k order.  Accept bits, momenta, proposals and
    positions are bit-identical to the oracle's f32-chain mode."""
    rho = 0.9
    fn_o = otargets.ar1_gaussian(rho, D)
    cov = otargets.ar1_covariance(rho, D)
    assert np.array_equal(cov, cov.T)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    q0 = prng.normal(prng.key(1), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.hmc(tgt, 0.5, dev_t(cov, dev), L, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev))
    np.testing.assert_array_equal(t2n(st_g.logdensity_grad), st_o.logdensity_grad)
    metric = _shared_factor(dev, cov, N, D)
    n_rej = 0
    for kk in prng.split(prng.key(0), 10):
        st_o, info_o = ohmc.kernel(kk, st_o, fn_o, np.float32(0.5), cov, L, chain_offset=3, metric=metric)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.momentum), info_o.momentum)
        assert np.array_equal(t2n(info_g.proposal.position), info_o.proposal.position)
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        assert np.array_equal(t2n(info_g.energy), info_o.energy)
        assert np.array_equal(t2n(info_g.acceptance_rate), info_o.acceptance_rate)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert np.array_equal(t2n(st_g.position), st_o.position)
        assert np.array_equal(t2n(st_g.logdensity_grad), st_o.logdensity_grad)
        assert 0.3 < info_o.acceptance_rate.mean() <= 1.0
        n_rej += int((~info_o.is_accepted).sum())
    assert n_rej > 0  # both branches of the select were exercised


def test_dense_f32chain_and_f64_oracle_modes_agree_to_roundoff(dev):
    """The order-independent fp64-accumulated oracle mode (what the per-chain / NUTS dense kernels
    match bit for bit) and the f32-chain mode differ by fp32 round-off only: one transition from the
    same state stays within 1e-4, so the choice of summation order is a rounding detail."""
    N, D, L = 64, 48, 6
    cov = otargets.ar1_covariance(0.9, D)
    fn_o = otargets.ar1_gaussian(0.9, D)
    st = ohmc.init(prng.normal(prng.key(4), (N, D)).astype(np.float32), fn_o)
    m32 = _shared_factor(dev, cov, N, D)
    m64 = m32._replace(dense_accum="f64")
    k = prng.key(6)
    _, i32 = ohmc.kernel(k, st, fn_o, np.float32(0.5), cov, L, metric=m32)
    _, i64 = ohmc.kernel(k, st, fn_o, np.float32(0.5), cov, L, metric=m64)
    np.testing.assert_allclose(i32.proposal.position, i64.proposal.position, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(i32.acceptance_rate, i64.acceptance_rate, rtol=1e-3, atol=1e-4)


def test_dense_hmc_statistics(dev):
    N, D, L = 2048, 16, 12
    rho = 0.9
    tgt = bjx.targets.AR1Gaussian(rho, D)
    cov = tgt.covariance(dev)
    alg = bjx.hmc(tgt, 0.6, cov, L)
    st = alg.init(torch.randn(N, D, device=dev))
    accs = []
    for kk in bjx.random.split(bjx.random.key(3), 25):
        st, info = alg.step(kk, st)
        accs.append(info.acceptance_rate.mean().item())
    x = st.position.double()
    emp = (x.T @ x) / N
    assert float((emp - cov.double()).abs().max()) < 0.15
    assert np.mean(accs) > 0.7


# ------------------------------------------------------------------ per-chain dense metric
def _random_spd(rng, N, D):
    a = rng.standard_normal((N, D, D))
    return (a @ np.swapaxes(a, 1, 2) / D + 0.5 * np.eye(D)).astype(np.float32)


@pytest.mark.parametrize("N,D", [(9, 7), (130, 64), (3, 200)])
def test_pc_matvec_t(dev, N, D):
    rng = np.random.default_rng(N + D)
    M = rng.standard_normal((N, D, D)).astype(np.float32)
    x = rng.standard_normal((N, D)).astype(np.float32)
    y = torch.empty(N, D, device=dev)
    Mt, xt = dev_t(M, dev), dev_t(x, dev)
    _lib.call("bjx_pc_matvec_t", _lib.current_stream(), N, D, Mt.data_ptr(), D * D, xt.data_ptr(),
              y.data_ptr())
    ref = np.einsum("nji,nj->ni", M.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert np.array_equal(t2n(y), ref)  # fp64 accumulation on both sides: bit-identical
    # one shared matrix applied to every chain (matrix_stride = 0)
    _lib.call("bjx_pc_matvec_t", _lib.current_stream(), N, D, Mt.data_ptr(), 0, xt.data_ptr(),
              y.data_ptr())
    ref0 = (x.astype(np.float64) @ M[0].astype(np.float64)).astype(np.float32)
    assert np.array_equal(t2n(y), ref0)


def test_welford_dense_kernels_bit_exact(dev):
    from blackjax_amd import adaptation as bad
    from oracle import adaptation as oad

    N, D = 11, 13
    rng = np.random.default_rng(5)
    wc_o = oad.welford_init(N, D, is_diag=False)
    wc = bad.WelfordAlgorithmState(torch.zeros(N, D, device=dev), torch.zeros(N, D, D, device=dev), 0)
    for _ in range(12):
        x = (rng.standard_normal((N, D)) * 2 + 1).astype(np.float32)
        wc_o = oad.welford_update(wc_o, x, is_diag=False)
        wc = bad._welford_update(wc, dev_t(x, dev))
    assert np.array_equal(t2n(wc.mean), wc_o.mean) and np.array_equal(t2n(wc.m2), wc_o.m2)
    for prev in (np.eye(D, dtype=np.float32), _random_spd(rng, N, D)):
        for shrink in (0.0, 2.0):
            prev_o = np.broadcast_to(prev, (N, D, D)) if prev.ndim == 2 else prev
            mm_o = oad.mm_final(oad.MassMatrixState(prev_o, wc_o), False, shrink)
            mm = bad._mm_final(bad.MassMatrixAdaptationState(dev_t(prev, dev), wc), shrink)
            assert np.array_equal(t2n(mm.inverse_mass_matrix), mm_o.inverse_mass_matrix)


def test_dense_pc_hmc_vs_oracle(dev):
    """One dense inverse mass matrix PER CHAIN (vmapped dense warmup output), per-chain step size,
    chain blocking.  fp64-accumulated matrix-vector products on both sides: the only non-bit-exact
    input is the fp64 Cholesky factor (LAPACK vs rocSOLVER), hence a 1e-6 tolerance."""
    N, D, L = 20, 12, 6
    rng = np.random.default_rng(3)
    imm = _random_spd(rng, N, D)
    eps = rng.uniform(0.1, 0.4, N).astype(np.float32)
    inv_var = rng.uniform(0.5, 2.0, D).astype(np.float32)
    fn_o = otargets.diag_gaussian(inv_var)
    q0 = prng.normal(prng.key(1), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.hmc(bjx.targets.DiagGaussian(dev_t(inv_var, dev)), dev_t(eps, dev), dev_t(imm, dev), L,
                  chain_block=8)
    st_g = alg.init(dev_t(q0, dev))
    for kk in prng.split(prng.key(0), 4):
        st_o, info_o = ohmc.kernel(kk, st_o, fn_o, eps, imm, L)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        np.testing.assert_allclose(t2n(info_g.momentum), info_o.momentum, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(t2n(st_g.position), st_o.position, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-4, atol=1e-6)


def test_dense_window_adaptation_vs_oracle(dev):
    """window_adaptation(hmc, is_mass_matrix_diagonal=False): dense Welford per chain
    (reference tests/mcmc/test_sampling.py:317-379 runs the same combination)."""
    from oracle import adaptation as oad

    N, D, L, T = 6, 5, 5, 60
    rho = 0.6
    fn_o = otargets.ar1_gaussian(rho, D)
    q0 = prng.normal(prng.key(4), (N, D)).astype(np.float32)
    st_o, par_o, hist_o = oad.window_adaptation_run(prng.key(7), q0, fn_o, T, L,
                                                    is_mass_matrix_diagonal=False)
    warm = bjx.window_adaptation(bjx.hmc, bjx.targets.AR1Gaussian(rho, D), is_mass_matrix_diagonal=False,
                                 num_integration_steps=L)
    (st_g, par_g), info = warm.run(prng.key(7), dev_t(q0, dev), T)
    assert par_g["inverse_mass_matrix"].shape == (N, D, D)
    eps_g = t2n(info.adaptation_state.step_size)
    sched = oad.build_schedule(T)
    t_end = [t for t, (_, e) in enumerate(sched) if e][0]  # first (only) window end
    for t in range(T):
        # bit-identical up to and including the window end (identity metric, dense Welford, blend);
        # afterwards the per-chain Cholesky factors come from different fp64 LAPACK back ends
        # (1e-16 relative), which the ill-conditioned 44-sample covariance and the freshly
        # re-initialised dual averaging amplify -- hence the looser bound on the last steps
        if t <= t_end:
            assert np.array_equal(eps_g[t], hist_o[t][1]), t
        else:
            np.testing.assert_allclose(eps_g[t], hist_o[t][1], rtol=2e-2)
    np.testing.assert_allclose(t2n(par_g["step_size"]), par_o["step_size"], rtol=2e-2)
    np.testing.assert_allclose(t2n(par_g["inverse_mass_matrix"]), par_o["inverse_mass_matrix"],
                               rtol=1e-5, atol=1e-6)
    # sampling with the adapted per-chain dense matrices runs
    alg = bjx.hmc(bjx.targets.AR1Gaussian(rho, D), par_g["step_size"], par_g["inverse_mass_matrix"], L)
    st, inf = alg.step(bjx.random.key(1), st_g)
    assert torch.isfinite(st.position).all() and inf.acceptance_rate.mean() > 0.3


def test_dense_full_size_properties(dev):
    """configs[4] at full size (16 384 x 512, AR(1) covariance as the dense inverse mass matrix):
    the MFMA leapfrog kernel against an fp64 torch evaluation on the whole batch, time reversibility
    of L velocity-Verlet steps and conservation of the Hamiltonian -- size-independent properties."""
    N, D, L, eps = 16384, 512, 8, 0.3
    rho = 0.9
    cov = torch.as_tensor(otargets.ar1_covariance(rho, D), device=dev)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    q0 = torch.randn(N, D, device=dev, generator=g)
    p0 = torch.randn(N, D, device=dev, generator=g) * 0.7
    logp0, g0 = tgt(q0)
    s = _lib.current_stream()

    # one fused kick + GEMM + drift launch vs fp64
    q1, p1 = torch.empty_like(q0), torch.empty_like(p0)
    _lib.call("bjx_leapfrog_dense", s, N, D, 1, eps, None, cov.data_ptr(), q0.data_ptr(), p0.data_ptr(),
              g0.data_ptr(), q1.data_ptr(), p1.data_ptr())
    p_ref = p0.double() + 0.5 * eps * g0.double()
    q_ref = q0.double() + eps * (p_ref @ cov.double().T)
    assert torch.equal(p1, torch.addcmul(p0, g0, torch.tensor(0.5 * eps, device=dev)).float()) or \
        (p1.double() - p_ref).abs().max().item() < 1e-6
    assert (q1.double() - q_ref).abs().max().item() < 2e-5 * q_ref.abs().max().item()

    def integrate(q, p, grad, steps):
        for i in range(steps):
            qn, pn = torch.empty_like(q), torch.empty_like(p)
            _lib.call("bjx_leapfrog_dense", s, N, D, 1 if i == 0 else 2, eps, None, cov.data_ptr(),
                      q.data_ptr(), p.data_ptr(), grad.data_ptr(), qn.data_ptr(), pn.data_ptr())
            q, p = qn, pn
            logp, grad = tgt(q)
        return q, p + (0.5 * eps) * grad, logp, grad

    def energy(logp, p):
        return -logp.double() + 0.5 * ((p.double() @ cov.double().T) * p.double()).sum(-1)

    qa, pa, logpa, ga = integrate(q0, p0, g0, L)
    h0, h1 = energy(logp0, p0), energy(logpa, pa)
    assert ((h1 - h0).abs() / h0.abs()).max().item() < 2e-2
    qb, pb, _, _ = integrate(qa, -pa, ga, L)
    assert (qb - q0).abs().max().item() < 5e-4
    assert (pb + p0).abs().max().item() < 5e-4


@pytest.mark.parametrize("N,D,L", [(100, 64, 4), (256, 128, 3), (200, 128, 3)])
def test_dense_hmc_reads_a_not_quite_symmetric_matrix_as_stored(dev, N, D, L):
    """A dense inverse mass matrix that is symmetric only up to rounding (what a dense Welford window produces):
    ``v = imm p`` must read it AS STORED, ``v[i] = sum_k imm[i][k] p[k]`` (metrics.py:263-304 ``linear_map``), on the
    whole-tile GEMM kernel AND on the general one (ragged N or D; round 4: they get the transposed copy) -- so a chain's
    arithmetic does not depend on the size of the batch it runs in.  Bit for bit against the oracle's f32-chain mode,
    which reads rows."""
    rho = 0.9
    fn_o = otargets.ar1_gaussian(rho, D)
    cov = otargets.ar1_covariance(rho, D).copy()
    rng = np.random.default_rng(3)
    cov = (cov * (1.0 + 1e-6 * np.triu(rng.standard_normal((D, D)), 1))).astype(np.float32)
    assert not np.array_equal(cov, cov.T)
    tgt = bjx.targets.AR1Gaussian(rho, D)
    q0 = prng.normal(prng.key(1), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.hmc(tgt, 0.5, dev_t(cov, dev), L)
    st_g = alg.init(dev_t(q0, dev))
    metric = _shared_factor(dev, cov, N, D)
    for kk in prng.split(prng.key(0), 3):
        st_o, info_o = ohmc.kernel(kk, st_o, fn_o, np.float32(0.5), cov, L, metric=metric)
        st_g, info_g = alg.step(kk, st_g)
        assert np.array_equal(t2n(info_g.proposal.position), info_o.proposal.position)
        assert np.array_equal(t2n(info_g.proposal.momentum), info_o.proposal.momentum)
        assert np.array_equal(t2n(info_g.energy), info_o.energy)
        assert np.array_equal(t2n(st_g.position), st_o.position)
