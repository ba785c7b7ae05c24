"""``blackjax_amd.targets.DeviceTarget``: a user-written HIP device log-density compiled at run time by hiprtc into
the engine's kernels (blackjax_amd/rtc.py, csrc/bjx_traj_dev.h).  The compile step needs no GPU (hiprtc
cross-compiles gfx950); loading and launching do (tests marked gpu)."""
import numpy as np
import pytest
import torch

import blackjax_amd as bjx

QUARTIC = r"""
// logp(q) = - sum_i ( a_i q_i^2 / 2 + c q_i^4 / 4 ),  params = [a_0 .. a_{D-1}, c]
struct Target {
  template <int NI> struct Ctx { F4 a[NI]; float c; };
  template <int NI> static __device__ void init(Ctx<NI>& ctx, int64_t D, const float* params) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) ctx.a[k] = ld4(params + j);
    }
    ctx.c = params[D];
  }
  template <int NI>
  static __device__ void eval(const Ctx<NI>& ctx, int64_t D, const float*, const F4 (&x)[NI], bool need_logp,
                              F4 (&g)[NI], float& lp) {
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int64_t j = ((int64_t)lane + 64 * k) * 4;
      if (j < D) {
        const float xs[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
        const float as[4] = {ctx.a[k].x, ctx.a[k].y, ctx.a[k].z, ctx.a[k].w};
        float gs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x2 = xs[e] * xs[e];
          gs[e] = -fmaf(ctx.c * x2, xs[e], as[e] * xs[e]);
          if (need_logp) acc += 0.5 * (double)as[e] * (double)x2 + 0.25 * (double)ctx.c * (double)x2 * (double)x2;
        }
        g[k] = F4{gs[0], gs[1], gs[2], gs[3]};
      }
    }
    if (need_logp) lp = (float)(-wave_sum(acc));
  }
};
"""


def test_device_target_source_compiles_for_gfx950_without_a_gpu():
    tgt = bjx.targets.DeviceTarget(QUARTIC)
    code = tgt.code_object()
    assert code[:4] == b"\x7fELF" and len(code) > 10000
    with pytest.raises(bjx.rtc.CompileError) as e:
        bjx.targets.DeviceTarget("struct Target { this is not C++ };").code_object()
    assert "error" in str(e.value)


def test_trajargs_mirror_has_the_size_the_header_asserts():
    import ctypes
    import re

    hdr = open(bjx.rtc.CSRC + "/bjx_traj_dev.h").read()
    assert int(re.search(r"sizeof\(TrajArgs\) == (\d+)", hdr).group(1)) == ctypes.sizeof(bjx.rtc.TrajArgs)


def _quartic(dev, D, c=0.3):
    g = torch.Generator(device=dev)
    g.manual_seed(D)
    a = (0.5 + torch.rand(D, device=dev, generator=g)).float()
    params = torch.cat([a, torch.tensor([c], device=dev)]).contiguous()
    ref = lambda q: (-(0.5 * a.double() * q.double() ** 2 + 0.25 * c * q.double() ** 4).sum(-1),  # noqa: E731
                     -(a.double() * q.double() + c * q.double() ** 3))
    return bjx.targets.DeviceTarget(QUARTIC, params), ref, a


@pytest.mark.gpu
@pytest.mark.parametrize("N,D", [(50, 256), (7, 320), (33, 1024), (5, 64)])
def test_device_target_callable_matches_a_float64_torch_reference(dev, N, D):
    tgt, ref, _ = _quartic(dev, D)
    q = torch.randn(N, D, device=dev)
    lp, g = tgt(q)
    lp_r, g_r = ref(q)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_r.float().cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(g.cpu().numpy(), g_r.float().cpu().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,L,per_chain", [(400, 256, 6, False), (60, 1024, 4, True), (31, 320, 3, False)])
def test_device_target_inside_the_trajectory_kernel_equals_the_external_callable_path(dev, N, D, L, per_chain):
    """The user's eval compiled INTO the whole-transition kernel gives the bits of the default path, where the
    same object is an external callable between two leapfrogs (and is recorded into the HIP graph driver)."""
    tgt, _, a = _quartic(dev, D)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    q0 = torch.randn(N, D, device=dev, generator=g)
    imm = (1.0 / a).contiguous()
    eps = 0.25
    if per_chain:
        imm = bjx.metrics.PerChainDiag((imm * (0.5 + torch.rand(N, D, device=dev, generator=g))).contiguous())
        eps = (0.25 * (0.5 + torch.rand(N, device=dev, generator=g))).contiguous()
    ref = bjx.hmc(tgt, eps, imm, L)
    fused = bjx.hmc(tgt, eps, imm, L, fuse_target=True)
    sa = sb = ref.init(q0)
    n_rej = 0
    for key in bjx.random.split(bjx.random.key(4), 3):
        sa, ia = ref.step(key, sa)
        sb, ib = fused.step(key, sb)
        for x, y in zip(sa, sb):
            assert torch.equal(x, y)
        for name in ("momentum", "acceptance_rate", "is_accepted", "is_divergent", "energy"):
            assert torch.equal(getattr(ia, name), getattr(ib, name)), name
        for x, y in zip(ia.proposal, ib.proposal):
            assert torch.equal(x, y)
        n_rej += int((~ia.is_accepted).sum())
    assert float(ia.acceptance_rate.mean()) > 0.5
    if N >= 400:
        assert n_rej > 0


@pytest.mark.gpu
def test_device_target_samples_its_density(dev):
    """c = 0: the quartic target is a Gaussian with variances 1 / a_i."""
    D = 256
    tgt, _, a = _quartic(dev, D, c=0.0)
    alg = bjx.hmc(tgt, 0.35, (1.0 / a).contiguous(), 7, fuse_target="lean")
    state = alg.init(torch.randn(4096, D, device=dev) / a.sqrt())
    for key in bjx.random.split(bjx.random.key(8), 60):
        state, info = alg.step(key, state)
    np.testing.assert_allclose(state.position.var(0).cpu().numpy(), (1.0 / a).cpu().numpy(), rtol=0.15)
    assert float(info.acceptance_rate.mean()) > 0.7


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,T", [(700, 256, 4), (90, 320, 5), (9000, 128, 3)])
def test_device_target_inside_the_nuts_tick_kernel_equals_the_external_callable_path(dev, N, D, T):
    """Free-running NUTS with the user's eval compiled into the multi-tick kernel of csrc/bjx_nuts_tick_dev.h (hiprtc
    compiles the library's own source file around it): positions, records and the final state are bit for bit
    those of the run where the same object is an external callable between two ticks."""
    tgt, _, a = _quartic(dev, D, c=0.6)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    q0 = torch.randn(N, D, device=dev, generator=g)
    alg = bjx.nuts(tgt, 0.35, (1.0 / a).contiguous(), max_num_doublings=6)
    st0 = alg.init(q0)
    key = bjx.random.key(13)
    st_a, pos_a, info_a = alg.run(key, st0, T)
    st_b, pos_b, info_b = alg.run(key, st0, T, fuse_target=True)
    assert torch.equal(pos_a, pos_b)
    for x, y in zip(st_a, st_b):
        assert torch.equal(x, y)
    for name in ("logdensity", "acceptance_rate", "energy", "num_integration_steps", "num_trajectory_expansions",
                 "is_divergent", "is_turning"):
        assert torch.equal(getattr(info_a, name), getattr(info_b, name)), name
    assert int(info_a.num_integration_steps.max()) > int(info_a.num_integration_steps.min())


@pytest.mark.gpu
def test_device_target_through_the_free_running_warmup_and_step(dev):
    """window_adaptation(nuts).run(free_running=True, fuse_target=True) and nuts(..., fuse_target=True).step with a
    user target: the per-chain adaptation code of csrc/bjx_nuts_tick_dev.h is part of the run-time kernel too."""
    N, D, T = 300, 256, 40
    tgt, _, a = _quartic(dev, D, c=0.6)
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    q0 = torch.randn(N, D, device=dev, generator=g)
    warm = bjx.window_adaptation(bjx.nuts, tgt, adaptation_info_fn=None, initial_step_size=0.3, max_num_doublings=6)
    (st_a, par_a), _ = warm.run(bjx.random.key(2), q0, T, free_running=True)
    (st_b, par_b), _ = warm.run(bjx.random.key(2), q0, T, free_running=True, fuse_target=True)
    assert torch.equal(st_a.position, st_b.position)
    assert torch.equal(par_a["step_size"], par_b["step_size"])
    assert torch.equal(par_a["inverse_mass_matrix"], par_b["inverse_mass_matrix"])
    imm = torch.ones(D, device=dev)
    ref, fused = bjx.nuts(tgt, 0.3, imm, max_num_doublings=6), bjx.nuts(tgt, 0.3, imm, max_num_doublings=6, fuse_target=True)
    sa, ia = ref.step(bjx.random.key(3), st_a)
    sb, ib = fused.step(bjx.random.key(3), st_a)
    assert torch.equal(sa.position, sb.position) and torch.equal(ia.num_integration_steps, ib.num_integration_steps)


def test_fuse_target_switches_validate_their_arguments_without_a_gpu():
    """Construction-time checks of the opt-in engine-resident switches (no device work involved)."""
    fn = lambda q: -0.5 * (q * q).sum(-1)  # noqa: E731
    with pytest.raises(NotImplementedError):
        bjx.hmc(fn, 0.1, torch.ones(256), 3, fuse_target=True, integrator=bjx.integrators.mclachlan)
    with pytest.raises(NotImplementedError):
        bjx.nuts(fn, 0.1, torch.ones(256), fuse_target=True, integrator=bjx.integrators.yoshida)
    with pytest.raises(NotImplementedError):
        bjx.window_adaptation(bjx.nuts, fn, fuse_target=True)
    with pytest.raises(NotImplementedError):
        bjx.window_adaptation(bjx.hmc, fn, fuse_target=True, is_mass_matrix_diagonal=False, num_integration_steps=3)
    with pytest.raises(ValueError):
        bjx.targets.DeviceTarget(QUARTIC, params=torch.ones(4, dtype=torch.float64))
    tgt = bjx.targets.DeviceTarget(QUARTIC)
    assert tgt._bjx_fused_target(256) == ("rtc", tgt) and tgt._bjx_fused_target(2048) is None
    assert bjx.rtc.nuts_kernel_name(256) == "bjx_rtc_nuts_multi_1_full" and bjx.rtc.nuts_kernel_name(320) == "bjx_rtc_nuts_multi_2"
    assert bjx.rtc.ni_for(132) == 1 and bjx.rtc.ni_for(512) == 2 and bjx.rtc.ni_for(1024) == 4


def test_nuts_translation_unit_compiles_for_gfx950_without_a_gpu():
    """hiprtc compiles csrc/bjx_nuts_tick_dev.h (the device functions the library is built from) around the user's struct: guards against a host
    include or host-only construct slipping into the device part of that file."""
    code = bjx.rtc.compile(bjx.rtc.NUTS_TU % {"source": QUARTIC, "struct": "Target"}, "nuts_user_test.hip")
    assert code[:4] == b"\x7fELF" and b"bjx_rtc_nuts_multi_1_full" in code
