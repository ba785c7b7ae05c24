"""The round-4 leaf kernels of the free-running NUTS ticks (k_nuts_async_tick3<GL, NI, WAVES>, rows of at most
256 floats: GL = 64 one chain per wave with half the registers of the v2 leaf -- the default of the busy phase
-- and GL = 16 four chains per wave, one 16-lane DPP row per chain) against the v2 leaf and the oracle.

The product takes the v3 leaf only for two-kernel ticks (more than BJX_NUTS_FUSED_ROWS = 8 192 live rows), so
the small shapes here run in a subprocess with BJX_NUTS_FUSED_ROWS=0 -- every tick of the run is then
[leaf kernel, transition-end kernel] -- with BJX_NUTS_LEAF3 = 68, 0 (the v2 leaf) and 16.  The kernels
promise the same bits (same expressions, same summation tree): every record and every position must be
IDENTICAL, and the v3 run must satisfy the oracle as the default path does.  The full C3 shape goes through
the v3 leaf by default: tests/test_full_shape_gpu.py::test_c3_*."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import blackjax_amd as bjx
    from oracle import prng
    dev = torch.device("cuda:0")
    out = {}

    def run_case(name, target, N, D, T, max_depth, eps, imm, q0, key_layout="step_major", chain_offset=0):
        alg = bjx.nuts(target, eps, imm, max_num_doublings=max_depth, chain_offset=chain_offset)
        st = alg.init(q0)
        final, pos, info = alg.run(bjx.random.key(42), st, T, key_layout=key_layout)
        out[name + ".pos"] = pos.cpu().numpy()
        out[name + ".final_g"] = final.logdensity_grad.cpu().numpy()
        for f in ("num_integration_steps", "num_trajectory_expansions", "is_divergent", "is_turning", "energy",
                  "acceptance_rate", "logdensity"):
            out[name + "." + f] = getattr(info, f).cpu().numpy()

    g = torch.Generator(device=dev); g.manual_seed(0)
    # > 256: two 16-byte pieces per lane; > 512: three / four (there the alternative is the round-1 general kernels)
    for D in (8, 64, 100, 128, 200, 256, 260, 384, 512, 640, 1024):
        N = 301 if D != 256 else 1030
        q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
        run_case(f"funnel{D}", bjx.targets.NealFunnel(), N, D, 5, 8, 0.1, torch.ones(D, device=dev), q0)
    # per-chain step sizes and per-chain diagonal metrics, chain-major keys, a chain offset
    N, D = 50, 36
    rng = np.random.default_rng(3)
    q0 = torch.as_tensor((0.5 * prng.normal(prng.key(2), (N, D))).astype(np.float32), device=dev)
    eps = torch.as_tensor(rng.uniform(0.05, 0.6, N).astype(np.float32), device=dev)
    imm = bjx.metrics.PerChainDiag(torch.as_tensor(rng.uniform(0.5, 2.0, (N, D)).astype(np.float32), device=dev))
    run_case("perchain", bjx.targets.NealFunnel(), N, D, 4, 5, eps, imm, q0, key_layout="chain_major", chain_offset=11)
    # diagonal Gaussian with the C2-style sigma ladder, a shared non-trivial metric
    N, D = 64, 256
    sig = (10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32)
    q0 = torch.as_tensor((prng.normal(prng.key(1), (N, D)) * sig).astype(np.float32), device=dev)
    run_case("gauss", bjx.targets.DiagGaussian(torch.as_tensor(1.0 / (sig * sig), device=dev)), N, D, 4, 6, 0.3,
             torch.as_tensor(sig * sig, device=dev), q0)
    # a multi-stage integrator: a leaf lasts three ticks (two middle stages + the closing tick)
    N, D = 70, 200
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    alg = bjx.nuts(bjx.targets.NealFunnel(), 0.15, torch.ones(D, device=dev), max_num_doublings=6,
                   integrator=bjx.integrators.yoshida)
    final, pos, info = alg.run(bjx.random.key(9), alg.init(q0), 3)
    out["yoshida.pos"] = pos.cpu().numpy()
    out["yoshida.n"] = info.num_integration_steps.cpu().numpy()
    out["yoshida.energy"] = info.energy.cpu().numpy()
    # free-running warm-up (per-chain adaptation inside the transition-end kernel)
    N, D = 96, 32
    q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
    warm = bjx.window_adaptation(bjx.nuts, bjx.targets.NealFunnel(), max_num_doublings=6)
    (st, par), info = warm.run(bjx.random.key(5), q0, 40, free_running=True)
    out["warm.pos"] = st.position.cpu().numpy()
    out["warm.eps"] = par["step_size"].cpu().numpy()
    out["warm.imm"] = torch.as_tensor(par["inverse_mass_matrix"]).cpu().numpy()
    np.savez(sys.argv[1], **out)
""") % ROOT


def _run(tmp_path, v3):
    path = str(tmp_path / f"v3_{v3}.npz")
    env = dict(os.environ, BJX_NUTS_FUSED_ROWS="0", BJX_NUTS_LEAF3=str(v3))
    r = subprocess.run([sys.executable, "-c", WORKER, path], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return dict(np.load(path))


def test_v3_leaf_is_bit_identical_to_v2_and_matches_the_oracle(tmp_path, dev):
    a = _run(tmp_path, 68)  # one chain per wave, lean register layout, work-list kernel for the transition ends
    b = _run(tmp_path, 0)   # the v2 leaf
    c = _run(tmp_path, 16)  # four chains per wave (kept as a measured alternative)
    d = _run(tmp_path, 132)  # the default of the busy phase: the same leaf, transition ends deferred into the next launch
    assert a.keys() == b.keys() == c.keys() == d.keys()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(c[k], b[k], equal_nan=True), k
        assert np.array_equal(d[k], b[k], equal_nan=True), k
    # trees of several depths, divergences and max-depth stops were in the comparison
    assert len(np.unique(a["funnel256.num_trajectory_expansions"])) >= 4
    assert a["perchain.num_trajectory_expansions"].max() == 5
    # and the v3 run satisfies the oracle (decisions exact, positions within the stated tolerance)
    from oracle import hmc as ohmc, nuts as onuts, prng, targets as otargets

    N, D, T = 50, 36, 4
    rng = np.random.default_rng(3)
    q0 = (0.5 * prng.normal(prng.key(2), (N, D))).astype(np.float32)
    eps = rng.uniform(0.05, 0.6, N).astype(np.float32)
    imm = rng.uniform(0.5, 2.0, (N, D)).astype(np.float32)
    fn_o = otargets.neal_funnel()
    st_o = ohmc.init(q0, fn_o)
    for t in range(T):
        ck = prng.split(prng.split(prng.key(42), N, offset=11), T)[:, t]
        st_o, info_o = onuts.kernel(None, st_o, fn_o, eps, imm, 5, chain_keys_override=ck, per_chain_diag=True)
        assert np.array_equal(a["perchain.num_integration_steps"][t], info_o.num_integration_steps), t
        assert np.array_equal(a["perchain.is_divergent"][t], info_o.is_divergent)
        assert np.array_equal(a["perchain.is_turning"][t], info_o.is_turning)
        np.testing.assert_allclose(a["perchain.pos"][t], st_o.position, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(a["perchain.energy"][t], info_o.energy, rtol=1e-6, atol=1e-6)
