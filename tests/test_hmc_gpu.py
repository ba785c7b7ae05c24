"""GPU parity: HIP HMC path (through the C ABI) vs the NumPy oracle on the same seeds.

Bar (BASELINE.json north_star): accept/reject decisions bit-exact; positions within a
stated fp tolerance.  With the shared numerics contract (explicit fma, fp64-accumulated
reductions) the element-wise state is in fact bit-identical; the tests
state ``ATOL_POS`` anyway and additionally count exact mismatches.
"""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import _lib
from oracle import hmc as ohmc
from oracle import prng, targets as otargets

pytestmark = pytest.mark.gpu

ATOL_POS = 1e-6  # stated tolerance on positions / momenta (relative to O(1) values)


def t2n(t):
    return t.detach().cpu().numpy()


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a), device=dev)


def sigma_ladder(D, lo=-1.0, hi=1.0):
    return (10.0 ** (lo + (hi - lo) * np.arange(D) / max(D - 1, 1))).astype(np.float32)


@pytest.mark.parametrize("N,D", [(7, 1024), (3, 100), (1, 5), (130, 64)])
def test_rng_normal_uniform(dev, N, D):
    key = prng.key(123)
    z = torch.empty(N, D, device=dev)
    u = torch.empty(N, device=dev)
    s = _lib.current_stream()
    _lib.call("bjx_rng_normal", s, int(key[0]), int(key[1]), 5, N, D, z.data_ptr())
    _lib.call("bjx_rng_uniform", s, int(key[0]), int(key[1]), 5, N, u.data_ptr())
    ck = prng.split(key, N, offset=5)
    z_ref = prng.normal(ck, (D,))
    u_ref = prng.uniform(ck, ())
    assert np.array_equal(t2n(u), u_ref)  # integer-derived: bit exact
    zz = t2n(z)
    mism = np.sum(zz != z_ref)
    assert mism <= max(1, zz.size // 100000), f"{mism} normal draws differ"
    np.testing.assert_allclose(zz, z_ref, rtol=2e-7, atol=0)


def test_device_log1p_exhaustive(dev):
    """csrc/bjx_log1p.h on the DEVICE, every fp32 t in (-1, 0]: the product's table-driven correctly-rounded -log1p
    equals the device library's fp64 log1p rounded once for all 1 065 353 217 inputs (the host run of the same source,
    tests/test_log1p_host.py, pins it on the C library's); the inputs its fast path defers are resolved by the
    generated slow table.  Arithmetic behind jax.random.normal (blackjax/util.py:88-91)."""
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    _lib.call("bjx_log1p_device_check", _lib.current_stream(), 1, counts.data_ptr())
    torch.cuda.synchronize()
    checked, bad, deferred, first_bad = (int(v) for v in counts.tolist())
    assert checked == 1065353217
    assert bad == 0, f"{bad} inputs differ, first t bits 0x{first_bad:08x}"
    assert 0 < deferred < 2000


def test_device_rng_matches_jax_docs_values(dev):
    """Row a34: the DEVICE threefry / uniform / erf_inv code (bjx_rng_key_probe: key used as is) against
    the six jax.random values printed in JAX's own documentation (tests/golden/reference_kats.json
    "jax_docs_streams": doc-sourced, no JAX on any box) -- and against the oracle on the same keys."""
    import json
    import os

    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["jax_docs_streams"]

    def probe(seed, D, n_children):
        k = prng.key(seed)
        z = torch.zeros(max(D, 1), device=dev)
        u = torch.zeros(1, device=dev)
        ch = torch.zeros(max(n_children, 1), 2, device=dev, dtype=torch.int32)
        _lib.call("bjx_rng_key_probe", _lib.current_stream(), int(k[0]), int(k[1]), D, z.data_ptr(), u.data_ptr(),
                  n_children, ch.data_ptr())
        return t2n(z)[:D], float(t2n(u)[0]), t2n(ch).view(np.uint32)[:n_children]

    z0, u0, c0 = probe(0, 3, 3)
    assert c0[:2].tolist() == d["split_key0"]
    assert c0[2].tolist() == d["legacy_split_key0_words"]
    assert abs(u0 - d["uniform_key0_scalar"]) < 5e-7 and np.float32(u0) == prng.uniform(prng.key(0))
    np.testing.assert_allclose(z0, np.float32(d["normal_key0_3"]), rtol=0, atol=2.4e-7)
    assert np.array_equal(z0, prng.normal(prng.key(0), (3,)))
    z42, _, c42 = probe(42, 1, 2)
    assert c42.tolist() == d["split_key42"]
    assert abs(float(z42[0]) - d["normal_key42_scalar"]) < 1e-8
    # a longer stream through the same probe against the oracle (counter i = element i)
    z, u, c = probe(2024, 4096, 100)
    assert np.array_equal(z, prng.normal(prng.key(2024), (4096,)))
    assert np.array_equal(c, prng.split(prng.key(2024), 100))


def test_rng_normal_large_sample_bit_exact(dev):
    """8.4 M normals: the device's fast correctly-rounded log1p path (bjx_log1p.h) against the
    oracle's fp64 log1p -- every draw bit-identical."""
    N, D = 2048, 4096
    key = prng.key(2024)
    z = torch.empty(N, D, device=dev)
    _lib.call("bjx_rng_normal", _lib.current_stream(), int(key[0]), int(key[1]), 0, N, D, z.data_ptr())
    z_ref = prng.normal(prng.split(key, N), (D,))
    zz = t2n(z)
    assert np.array_equal(zz, z_ref), f"{np.sum(zz != z_ref)} of {zz.size} normal draws differ"
    assert abs(float(zz.mean())) < 2e-3 and abs(float(zz.std()) - 1.0) < 2e-3


@pytest.mark.parametrize("N,D,per_chain_imm", [(64, 1024, False), (5, 100, False), (9, 256, True), (3, 7, True),
                                               (37, 64, False), (11, 16, True), (6, 128, True), (1, 4, False), (70, 32, True), (19, 8, False)])
def test_momentum_diag(dev, N, D, per_chain_imm):
    rng = np.random.default_rng(0)
    imm = rng.uniform(0.1, 4.0, size=(N, D) if per_chain_imm else (D,)).astype(np.float32)
    key = prng.key(7)
    p = torch.empty(N, D, device=dev)
    ke = torch.empty(N, device=dev)
    immt = dev_t(imm, dev)
    _lib.call("bjx_hmc_momentum_diag", _lib.current_stream(), int(key[0]), int(key[1]), 11, -1, N, D,
              immt.data_ptr(), D if per_chain_imm else 0, p.data_ptr(), ke.data_ptr())
    metric = ohmc.default_metric(imm, n_chains=N)
    kk = prng.split(prng.split(key, N, offset=11), 2)
    p_ref = ohmc.sample_momentum(metric, kk[:, 0], D)
    ke_ref = ohmc.kinetic_energy(metric, p_ref)
    pp = t2n(p)
    assert np.sum(pp != p_ref) <= max(1, pp.size // 100000)
    np.testing.assert_allclose(pp, p_ref, rtol=2e-7)
    np.testing.assert_allclose(t2n(ke), ke_ref, rtol=1e-6)


@pytest.mark.parametrize("N,D", [(33, 1024), (4, 6), (2, 1023), (5, 2048), (3, 3072)])
@pytest.mark.parametrize("per_chain", [False, True])
def test_leapfrog_diag_bit_exact(dev, N, D, per_chain):
    rng = np.random.default_rng(1)
    q = rng.standard_normal((N, D)).astype(np.float32)
    p = rng.standard_normal((N, D)).astype(np.float32)
    imm = rng.uniform(0.1, 4.0, size=(N, D) if per_chain else (D,)).astype(np.float32)
    eps = rng.uniform(0.01, 0.3, size=N).astype(np.float32) if per_chain else np.float32(0.1)
    fn = otargets.diag_gaussian(np.ones(D, np.float32))
    logp, g = fn(q)
    metric = ohmc.default_metric(imm, n_chains=N)
    z = ohmc.IntegratorState(q, p, logp, g)
    z1 = ohmc.velocity_verlet(z, eps, fn, metric)      # p_half+drift, grad, closing kick
    z2 = ohmc.velocity_verlet(z1, eps, fn, metric)

    qt, pt, gt, immt = (dev_t(a, dev) for a in (q, p, g, imm))
    eps_pc = dev_t(eps, dev) if per_chain else None
    s = _lib.current_stream()
    qo, po = torch.empty_like(qt), torch.empty_like(pt)
    _lib.call("bjx_leapfrog_diag", s, N, D, 1, 0.0 if per_chain else float(eps), _lib.ptr(eps_pc),
              immt.data_ptr(), D if per_chain else 0, qt.data_ptr(), pt.data_ptr(), gt.data_ptr(),
              qo.data_ptr(), po.data_ptr())
    assert np.array_equal(t2n(qo), z1.position)
    g1 = dev_t(z1.logdensity_grad, dev)
    # second step, in place, two kicks (closing kick of step 1 + opening kick of step 2)
    _lib.call("bjx_leapfrog_diag", s, N, D, 2, 0.0 if per_chain else float(eps), _lib.ptr(eps_pc),
              immt.data_ptr(), D if per_chain else 0, qo.data_ptr(), po.data_ptr(), g1.data_ptr(),
              qo.data_ptr(), po.data_ptr())
    assert np.array_equal(t2n(qo), z2.position)


@pytest.mark.parametrize("N,D", [(9, 2048), (6, 1024), (5, 96)])
def test_leapfrog_diag_masked_matches_unmasked(dev, N, D):
    """dynamic HMC's per-chain trajectory lengths (mcmc/dynamic_hmc.py): chains whose trajectory is
    complete (step_idx >= n_steps) are left untouched in place and copied through out of place; the
    others get exactly the unmasked stage.  D = 2048 / 1024 take the flat kernel, 96 the row kernel."""
    g_ = torch.Generator(device=dev)
    g_.manual_seed(3)
    q, p, g = (torch.randn(N, D, device=dev, generator=g_) for _ in range(3))
    imm = torch.rand(N, D, device=dev, generator=g_) + 0.5
    eps = torch.rand(N, device=dev, generator=g_) * 0.2 + 0.01
    n_steps = torch.tensor([(i * 2) % 5 for i in range(N)], dtype=torch.int32, device=dev)
    step_idx = 2
    s = _lib.current_stream()
    q_ref, p_ref = torch.empty_like(q), torch.empty_like(p)
    _lib.call("bjx_leapfrog_diag", s, N, D, 2, 0.0, eps.data_ptr(), imm.data_ptr(), D, q.data_ptr(),
              p.data_ptr(), g.data_ptr(), q_ref.data_ptr(), p_ref.data_ptr())
    live = (n_steps > step_idx)[:, None]
    want_q, want_p = torch.where(live, q_ref, q), torch.where(live, p_ref, p)
    q_out, p_out = torch.full_like(q, float("nan")), torch.full_like(p, float("nan"))
    _lib.call("bjx_leapfrog_diag_masked", s, N, D, 2, 0.0, eps.data_ptr(), imm.data_ptr(), D, q.data_ptr(),
              p.data_ptr(), g.data_ptr(), q_out.data_ptr(), p_out.data_ptr(), n_steps.data_ptr(), step_idx)
    assert torch.equal(q_out, want_q) and torch.equal(p_out, want_p)
    q_in, p_in = q.clone(), p.clone()
    _lib.call("bjx_leapfrog_diag_masked", s, N, D, 2, 0.0, eps.data_ptr(), imm.data_ptr(), D, q_in.data_ptr(),
              p_in.data_ptr(), g.data_ptr(), q_in.data_ptr(), p_in.data_ptr(), n_steps.data_ptr(), step_idx)
    assert torch.equal(q_in, want_q) and torch.equal(p_in, want_p)


def _run_both(dev, N, D, L, T, eps, imm, inv_var, seed=0, chain_offset=0, q_scale=1.0):
    fn_o = otargets.diag_gaussian(inv_var)
    fn_g = bjx.targets.DiagGaussian(dev_t(inv_var, dev))
    q0 = (q_scale * prng.normal(prng.key(1), (N, D))).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    alg = bjx.hmc(fn_g, eps if np.ndim(eps) == 0 else dev_t(eps, dev), dev_t(imm, dev), L,
                  chain_offset=chain_offset)
    st_g = alg.init(dev_t(q0, dev))
    assert np.array_equal(t2n(st_g.logdensity), st_o.logdensity)
    keys = prng.split(prng.key(seed), T)
    out = []
    for t in range(T):
        st_o, info_o = ohmc.kernel(keys[t], st_o, fn_o, eps, imm, L, chain_offset=chain_offset)
        st_g, info_g = alg.step(keys[t], st_g)
        out.append((st_o, info_o, st_g, info_g))
    return out


def _check_step(st_o, info_o, st_g, info_g):
    assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted), "accept/reject differs"
    assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
    np.testing.assert_allclose(t2n(st_g.position), st_o.position, atol=ATOL_POS, rtol=ATOL_POS)
    np.testing.assert_allclose(t2n(info_g.momentum), info_o.momentum, atol=ATOL_POS, rtol=ATOL_POS)
    np.testing.assert_allclose(t2n(info_g.proposal.momentum), info_o.proposal.momentum,
                               atol=ATOL_POS, rtol=ATOL_POS)
    np.testing.assert_allclose(t2n(info_g.acceptance_rate), info_o.acceptance_rate, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(t2n(info_g.energy), info_o.energy, rtol=1e-6)
    np.testing.assert_allclose(t2n(st_g.logdensity), st_o.logdensity, rtol=1e-6)
    np.testing.assert_allclose(t2n(st_g.logdensity_grad), st_o.logdensity_grad, atol=ATOL_POS, rtol=ATOL_POS)
    assert info_g.num_integration_steps == info_o.num_integration_steps


def test_hmc_config1_parity(dev):
    """BASELINE.json configs[0]: 1024-dim isotropic Gaussian, 128 chains, 10 leapfrog steps."""
    N, D, L, T = 128, 1024, 10, 20
    res = _run_both(dev, N, D, L, T, np.float32(0.1), np.ones(D, np.float32), np.ones(D, np.float32))
    n_acc = 0
    exact = True
    for st_o, info_o, st_g, info_g in res:
        _check_step(st_o, info_o, st_g, info_g)
        n_acc += int(info_o.is_accepted.sum())
        exact &= np.array_equal(t2n(st_g.position), st_o.position)
    assert 0 < n_acc < N * T  # both branches of the Metropolis select were exercised
    assert exact, "positions are expected to be bit-identical under the shared numerics contract"


def test_hmc_diag_ladder_parity(dev):
    """Scaled-down configs[1]: diagonal Gaussian sigma ladder, imm = sigma^2, eps 0.25, L=50."""
    N, D, L, T = 96, 256, 50, 4
    sig = sigma_ladder(D)
    imm = (sig * sig).astype(np.float32)
    inv_var = (np.float32(1.0) / imm).astype(np.float32)
    res = _run_both(dev, N, D, L, T, np.float32(0.25), imm, inv_var, seed=3, chain_offset=1000,
                    q_scale=1.0)
    for r in res:
        _check_step(*r)


def test_hmc_ragged_and_per_chain(dev):
    """D not a multiple of 4 (scalar path), N=1..3, per-chain step size and per-chain imm."""
    for N, D in [(1, 5), (3, 101), (2, 1)]:
        rng = np.random.default_rng(N * 100 + D)
        imm = rng.uniform(0.5, 2.0, size=(N, D)).astype(np.float32) if N != D else rng.uniform(0.5, 2.0, size=(D,)).astype(np.float32)
        eps = rng.uniform(0.05, 0.4, size=N).astype(np.float32)
        inv_var = rng.uniform(0.5, 2.0, size=D).astype(np.float32)
        res = _run_both(dev, N, D, 7, 6, eps, imm, inv_var, seed=5)
        for r in res:
            _check_step(*r)


def test_hmc_divergence_and_nan(dev):
    """Huge step size -> energy error > threshold -> is_divergent and rejection; NaN energy -> reject
    (proposal.py:45-48, hmc.py:162)."""
    N, D = 8, 64
    res = _run_both(dev, N, D, 20, 2, np.float32(50.0), np.ones(D, np.float32), np.ones(D, np.float32))
    for st_o, info_o, st_g, info_g in res:
        assert np.array_equal(t2n(info_g.is_accepted), info_o.is_accepted)
        assert np.array_equal(t2n(info_g.is_divergent), info_o.is_divergent)
        assert info_o.is_divergent.all() and not info_o.is_accepted.any()
        assert np.array_equal(t2n(st_g.position), st_o.position)


def test_autograd_logdensity_callable(dev):
    """A plain torch log-density (no gradient returned) goes through torch.autograd."""
    N, D = 16, 32
    fn = lambda q: -0.5 * (q * q).sum(-1)
    alg = bjx.hmc(fn, 0.2, torch.ones(D, device=dev), 5)
    st = alg.init(torch.randn(N, D, device=dev))
    st2, info = alg.step(bjx.random.key(0), st)
    assert st2.position.shape == (N, D) and info.acceptance_rate.shape == (N,)
    assert torch.isfinite(st2.position).all()
    assert (info.acceptance_rate > 0.5).all()


def test_cpu_tensor_rejected():
    with pytest.raises(RuntimeError):
        bjx.hmc.init(torch.zeros(2, 3), lambda q: -0.5 * (q * q).sum(-1))


def test_moments_full_size_property(dev):
    """Size-independent property at a large size: on N(0, sigma^2) with ideal mass, the pooled
    second moment over many chains matches sigma^2 and acceptance is healthy."""
    N, D, L = 8192, 256, 10
    sig = sigma_ladder(D)
    imm = dev_t(sig * sig, dev)
    fn = bjx.targets.DiagGaussian(dev_t(1.0 / (sig * sig), dev))
    alg = bjx.hmc(fn, 0.25, imm, L)
    st = alg.init(dev_t(sig, dev) * torch.randn(N, D, device=dev))
    keys = bjx.random.split(bjx.random.key(9), 12)
    accs = []
    for k in keys:
        st, info = alg.step(k, st)
        accs.append(info.acceptance_rate.mean().item())
    var = (st.position ** 2).mean(0).cpu().numpy()
    np.testing.assert_allclose(var, sig * sig, rtol=0.1)
    assert np.mean(accs) > 0.7


def test_chain_block_tiling_is_invisible(dev):
    """Running a transition block-by-block over chains (Infinity-Cache tiling) gives identical
    results, including HMCInfo, because per-chain keys depend only on the global chain index."""
    N, D, L = 200, 128, 9
    sig = sigma_ladder(D)
    imm = dev_t(sig * sig, dev)
    fn = bjx.targets.DiagGaussian(dev_t(1.0 / (sig * sig), dev))
    q0 = dev_t(sig, dev) * torch.randn(N, D, device=dev)
    eps = torch.rand(N, device=dev) * 0.3 + 0.05
    a1 = bjx.hmc(fn, eps, imm, L)
    a2 = bjx.hmc(fn, eps, imm, L, chain_block=48)
    a3 = bjx.hmc(fn, eps, imm, L, chain_block=64, use_graph=True)  # HIP-graph captured inner loop
    a4 = bjx.hmc(fn, eps, imm, L, use_graph=True)
    s1, s2, s3, s4 = a1.init(q0), a2.init(q0), a3.init(q0), a4.init(q0)
    for k in bjx.random.split(bjx.random.key(4), 3):
        s1, i1 = a1.step(k, s1)
        s2, i2 = a2.step(k, s2)
        s3, i3 = a3.step(k, s3)
        s4, i4 = a4.step(k, s4)
        for sx, ix in ((s2, i2), (s3, i3), (s4, i4)):
            for x, y in zip(s1, sx):
                assert torch.equal(x, y)
            for x, y in zip(i1[:5], ix[:5]):
                assert torch.equal(x, y)
            for x, y in zip(i1.proposal, ix.proposal):
                assert torch.equal(x, y)


# ------------------------------------------------------------------ full-size (BASELINE.json configs[1]) properties
FULL_N, FULL_D = 65536, 1024


def _full_setup(dev):
    sig = sigma_ladder(FULL_D)
    imm = dev_t(sig * sig, dev)
    fn = bjx.targets.DiagGaussian(dev_t(1.0 / (sig * sig), dev))
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    q0 = dev_t(sig, dev) * torch.randn(FULL_N, FULL_D, device=dev, generator=g)
    return sig, imm, fn, q0


def test_full_size_sharding_invariance_and_determinism(dev):
    """At 65 536 x 1 024: (a) the same seed twice gives bit-identical output; (b) running the chain
    range as two shards with chain_offset (what two GPUs would do) reproduces the unsharded
    transition bit for bit -- results do not depend on the number of ranks."""
    sig, imm, fn, q0 = _full_setup(dev)
    L = 5
    key = bjx.random.key(2024)
    alg = bjx.hmc(fn, 0.25, imm, L)
    st0 = alg.init(q0)
    s1, i1 = alg.step(key, st0)
    s2, i2 = alg.step(key, st0)
    assert torch.equal(s1.position, s2.position) and torch.equal(i1.is_accepted, i2.is_accepted)
    half = FULL_N // 2
    outs = []
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        a = bjx.hmc(fn, 0.25, imm, L, chain_offset=r * half)
        sr, ir = a.step(key, bjx.hmc.init(q0[sl].contiguous(), fn))
        outs.append((sr, ir))
    assert torch.equal(torch.cat([o[0].position for o in outs]), s1.position)
    assert torch.equal(torch.cat([o[1].is_accepted for o in outs]), i1.is_accepted)
    assert torch.equal(torch.cat([o[1].acceptance_rate for o in outs]), i1.acceptance_rate)
    acc = i1.acceptance_rate.mean().item()
    assert 0.5 < acc <= 1.0 + 1e-5


def test_full_size_leapfrog_reversibility_and_energy(dev):
    """Size-independent integrator properties at 65 536 x 1 024: integrating L steps, flipping the
    momentum and integrating L more steps returns to the start (velocity-Verlet is time
    reversible) and the Hamiltonian is conserved to O(eps^2)."""
    sig, imm, fn, q0 = _full_setup(dev)
    N, D, L, eps = FULL_N, FULL_D, 10, 0.25
    s = _lib.current_stream()
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    p0 = torch.randn(N, D, device=dev, generator=g) / dev_t(sig, dev)
    logp0, grad = fn(q0)
    h0 = -logp0 + 0.5 * (imm * p0 * p0).sum(-1)

    def integrate(q, p, grad, steps):
        q, p = q.clone(), p.clone()
        for i in range(steps):
            _lib.call("bjx_leapfrog_diag", s, N, D, 1 if i == 0 else 2, eps, None, imm.data_ptr(), 0,
                      q.data_ptr(), p.data_ptr(), grad.data_ptr(), q.data_ptr(), p.data_ptr())
            logp, grad = fn(q)
        p = p + (eps * 0.5) * grad  # closing half kick
        return q, p, logp, grad

    q1, p1, logp1, g1 = integrate(q0, p0, grad, L)
    h1 = -logp1 + 0.5 * (imm * p1 * p1).sum(-1)
    rel = ((h1 - h0).abs() / h0.abs()).max().item()
    assert rel < 5e-3
    q2, p2, _, _ = integrate(q1, -p1, g1, L)
    scale = dev_t(sig, dev)
    assert ((q2 - q0).abs() / scale).max().item() < 2e-4
    assert ((p2 + p0).abs() * scale).max().item() < 2e-4


def test_hmc_default_driver_graph_and_fallback(dev):
    """use_graph="auto" (default): for a callable declared recordable (blackjax_amd.capturable) small
    blocks are driven through a HIP graph of the inner loop; a callable that then turns out to
    synchronise with the host falls back to plain launches; an undeclared callable is never
    recorded -- same results in all cases."""
    N, D, L = 200, 16, 7
    g_ = torch.Generator(device=dev)
    g_.manual_seed(9)
    q0 = torch.randn(N, D, device=dev, generator=g_)
    iv = torch.linspace(0.5, 2.0, D, device=dev)

    def plain(q):
        return -0.5 * (q * q * iv).sum(-1)

    def syncing(q):
        lp = -0.5 * (q * q * iv).sum(-1)
        assert float(lp.sum().item()) < float("inf")  # host sync: cannot be recorded
        return lp

    def undeclared(q):
        return -0.5 * (q * q * iv).sum(-1)

    # the three are compared bit for bit, and `syncing` cannot be traced into a generated kernel (it reads a value on the
    # host): keep them all on eager autograd (the traced form accumulates logp in fp64: other last bits)
    for f in (plain, syncing, undeclared):
        bjx.no_trace(f)
    imm = torch.ones(D, device=dev)
    ref = bjx.hmc(plain, 0.2, imm, L, use_graph=False)
    st = ref.init(q0)
    keys = prng.split(prng.key(6), 3)

    for fn in (bjx.capturable(plain), bjx.capturable(syncing), undeclared):
        alg = bjx.hmc(fn, 0.2, imm, L)  # default driver
        s_r, s_a = st, st
        for k in keys:
            s_r, i_r = ref.step(k, s_r)
            s_a, i_a = alg.step(k, s_a)
            assert torch.equal(s_r.position, s_a.position)
            assert torch.equal(i_r.acceptance_rate, i_a.acceptance_rate)
            assert torch.equal(i_r.proposal.position, i_a.proposal.position)


_SWEEP = [(n, d, l, pc) for (n, d, l, pc) in [
    (7, 4, 3, False), (19, 8, 2, True), (33, 12, 4, False), (65, 16, 3, True), (9, 20, 5, True),
    (130, 32, 2, False), (5, 36, 3, True), (70, 64, 4, True), (3, 68, 2, False), (41, 100, 3, True),
    (17, 128, 2, False), (6, 132, 3, True), (11, 200, 2, False), (4, 256, 3, True), (13, 300, 2, True),
    (2, 1000, 2, False), (5, 1024, 3, True), (3, 1028, 2, True), (2, 2048, 2, False), (1, 5, 2, True),
    (8, 7, 3, False), (6, 63, 2, True), (3, 1023, 2, False), (257, 48, 2, True)]]


@pytest.mark.parametrize("N,D,L,per_chain", _SWEEP)
def test_hmc_shape_sweep(dev, N, D, L, per_chain):
    """Every row-length variant of the diagonal-metric kernels (scalar rows, 4 ... 32 lanes per row, flat
    leapfrog for rows that do not fill whole waves, one wave per row, flat rows of a multiple of 1 024)
    against the oracle, with shared and per-chain step sizes / metrics: accept bits and positions."""
    rng = np.random.default_rng(N * 1000 + D)
    inv_var = rng.uniform(0.5, 2.0, D).astype(np.float32)
    if per_chain:
        imm = rng.uniform(0.5, 2.0, (N, D)).astype(np.float32)
        eps = rng.uniform(0.05, 0.3, N).astype(np.float32)
    else:
        imm = rng.uniform(0.5, 2.0, D).astype(np.float32)
        eps = np.float32(0.2)
    fn_o = otargets.diag_gaussian(inv_var)
    fn_g = bjx.targets.DiagGaussian(dev_t(inv_var, dev))
    q0 = prng.normal(prng.key(D), (N, D)).astype(np.float32)
    st_o = ohmc.init(q0, fn_o)
    imm_g = bjx.metrics.PerChainDiag(dev_t(imm, dev)) if per_chain else dev_t(imm, dev)
    alg = bjx.hmc(fn_g, dev_t(eps, dev) if per_chain else float(eps), imm_g, L, chain_offset=3)
    st_g = alg.init(dev_t(q0, dev))
    for k in prng.split(prng.key(N), 3):
        st_o, info_o = ohmc.kernel(k, st_o, fn_o, eps, imm, L, chain_offset=3)
        st_g, info_g = alg.step(k, st_g)
        _check_step(st_o, info_o, st_g, info_g)
        assert np.array_equal(t2n(st_g.position), st_o.position)
