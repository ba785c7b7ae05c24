"""``blackjax_amd.targets.from_elementwise``: a plain PyTorch log-density of the element-wise +
row-sum shape is traced with torch.fx, differentiated in forward mode and emitted as ONE HIP value-and-gradient kernel
(the reference gets this fusion from jax.value_and_grad under XLA: blackjax/mcmc/integrators.py:189,204).
CPU part: tracing, code generation, the hiprtc cross-compile, and what is refused; the GPU part compares the
generated kernel with torch.autograd on the same function."""
import math

import numpy as np
import pytest
import torch

import blackjax_amd as bjx
from blackjax_amd import elementwise as ew

D = 256
_g = torch.Generator().manual_seed(0)
IV = torch.rand(D, generator=_g) + 0.5
MU = torch.randn(D, generator=_g)

FUNCTIONS = {
    "gaussian": lambda q: -0.5 * (q * q * IV).sum(-1),
    "readme": lambda q: -0.5 * (q * q).sum(-1),
    "shifted_normalised": lambda q: -0.5 * (((q - MU) ** 2) * IV).sum(-1) - 0.5 * D * math.log(2 * math.pi),
    "logistic_plus_prior": lambda q: (-torch.nn.functional.softplus(-q * MU)).sum(-1) - 0.5 * (q ** 2).sum(-1) / 4.0,
    "student_t": lambda q: (-2.5 * torch.log1p(q * q / 4.0)).sum(dim=-1),
    "mixed": lambda q: torch.sum(torch.tanh(q) * IV - torch.exp(-q.abs()) + torch.sigmoid(q) / (1.0 + q * q), -1),
}


@pytest.mark.parametrize("name", sorted(FUNCTIONS))
def test_trace_generates_source_that_hiprtc_compiles(name):
    src = ew.trace(FUNCTIONS[name], D)
    assert "struct Target" in src.source and src.n_terms >= 1
    code = bjx.rtc.compile(bjx.rtc.TARGET_TU % {"source": src.source, "struct": "Target"}, f"ew_{name}.hip")
    assert code[:4] == b"\x7fELF"


def test_what_is_refused_says_why():
    for bad, word in ((lambda q: torch.logsumexp(q, -1), "unsupported function"),
                      (lambda q: (q @ torch.eye(D)).sum(-1), "shape"),
                      (lambda q: torch.exp((q * q).sum(-1)), "non-linear"),
                      (lambda q: (q * q).sum(0), "last axis"),
                      (lambda q: (q * q).sum(-1, True), "last axis"),          # positional keepdim: (N, 1), not the row sum
                      (lambda q: torch.sum(q * q, -1, True), "last axis"),
                      (lambda q: (q * q).sum(-1, keepdim=True), "last axis"),
                      (lambda q: (q * q).sum(-1, dtype=torch.float64), "last axis"),
                      (lambda q: q * q, "sum over the last axis")):
        with pytest.raises(NotImplementedError, match=word):
            ew.trace(bad, D)
    with pytest.raises(NotImplementedError, match="could not trace"):
        ew.trace(lambda q: (q * q).sum(-1) if q.sum() > 0 else q.sum(-1), D)  # data-dependent control flow


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FUNCTIONS))
@pytest.mark.parametrize("N,Dg", [(37, 256), (300, 1024), (16, 64)])
def test_generated_kernel_matches_autograd(name, N, Dg):
    """Stated tolerance: every traced op is one fp32 operation as in eager PyTorch, the derivative is the forward-mode
    formula and the row sum is fp64-accumulated: gradient within 2e-6 relative (+ 1e-6 absolute), logp within 2e-6
    relative of autograd's."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    iv = (torch.rand(Dg, device=dev, generator=g) + 0.5)
    mu = torch.randn(Dg, device=dev, generator=g)
    fns = {
        "gaussian": lambda q: -0.5 * (q * q * iv).sum(-1),
        "readme": lambda q: -0.5 * (q * q).sum(-1),
        "shifted_normalised": lambda q: -0.5 * (((q - mu) ** 2) * iv).sum(-1) - 0.5 * Dg * math.log(2 * math.pi),
        "logistic_plus_prior": lambda q: (-torch.nn.functional.softplus(-q * mu)).sum(-1) - 0.5 * (q ** 2).sum(-1) / 4.0,
        "student_t": lambda q: (-2.5 * torch.log1p(q * q / 4.0)).sum(dim=-1),
        "mixed": lambda q: torch.sum(torch.tanh(q) * iv - torch.exp(-q.abs()) + torch.sigmoid(q) / (1.0 + q * q), -1),
    }
    fn = fns[name]
    tgt = bjx.targets.from_elementwise(fn, Dg, device=dev)
    q = 1.5 * torch.randn(N, Dg, device=dev, generator=g)
    lp, grad = tgt(q)
    qa = q.clone().requires_grad_(True)
    lp_a = fn(qa)
    (g_a,) = torch.autograd.grad(lp_a.sum(), qa)
    # the reference for the tolerance: the same function in float64
    qd = q.double().requires_grad_(True)
    np.testing.assert_allclose(grad.cpu().numpy(), g_a.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_a.detach().cpu().numpy(), rtol=2e-5, atol=1e-4)
    del qd


@pytest.mark.gpu
def test_hmc_on_a_traced_callable_equals_hmc_on_the_library_gaussian():
    """The traced README Gaussian as the external callable of blackjax_amd.hmc: same accept decisions as the library's
    DiagGaussian target on the same keys (gradients agree to rounding, so do positions)."""
    dev = torch.device("cuda:0")
    N, Dg, L = 512, 1024, 8
    sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(Dg) / (Dg - 1))).astype(np.float32), device=dev)
    inv_var = (1.0 / (sig * sig)).contiguous()
    imm = (sig * sig).contiguous()
    fn = lambda q: -0.5 * (q * q * inv_var).sum(-1)  # noqa: E731
    traced = bjx.targets.from_elementwise(fn, Dg, device=dev)
    a1 = bjx.hmc(traced, 0.25, imm, L)
    a2 = bjx.hmc(bjx.targets.DiagGaussian(inv_var), 0.25, imm, L)
    q0 = sig * torch.randn(N, Dg, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    s1, s2 = a1.init(q0), a2.init(q0)
    same = 0
    for k in bjx.random.split(bjx.random.key(0), 4):
        s1, i1 = a1.step(k, s1)
        s2, i2 = a2.step(k, s2)
        same += int((i1.is_accepted == i2.is_accepted).sum())
        s1 = s2  # keep the two on the same trajectory: only rounding separates them within a transition
        np.testing.assert_allclose(i1.acceptance_rate.cpu().numpy(), i2.acceptance_rate.cpu().numpy(), atol=2e-3)
    assert same >= 4 * N - 4


def test_any_row_length_cross_compiles_without_a_gpu():
    """D > 1 024 and D % 4 != 0 take the row-loop form of the same generated arithmetic (ElementwiseRowsTarget); D <= 1 024,
    D % 4 == 0 stays a DeviceTarget (registers, usable with fuse_target)."""
    assert type(bjx.targets.from_elementwise(FUNCTIONS["readme"], 1024, device="cpu")).__name__ == "DeviceTarget"
    for d in (4096, 1003, 2050, 1):
        fn = (lambda q: -0.5 * (q * q).sum(-1)) if d != 1003 else (lambda q: (-2.5 * torch.log1p(q * q / 4.0)).sum(dim=-1))
        t = bjx.targets.from_elementwise(fn, d, device="cpu")
        assert type(t).__name__ == "ElementwiseRowsTarget" and t._bjx_fused_target(d) is None
        assert "bjx_rtc_ew_rows" in t.elementwise.rows_source and t.code_object()[:4] == b"\x7fELF"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FUNCTIONS))
@pytest.mark.parametrize("N,Dg", [(19, 4096), (7, 1003), (5, 2050), (64, 1), (300, 1024)])
def test_row_loop_kernel_matches_autograd_and_the_register_form(name, N, Dg):
    """BASELINE.json configs[3] has D = 4 096: the row-loop kernel serves it.  Same tolerance as
    above; for the C2 Gaussian (`gaussian`: three fp32 multiplications per element and an fp64 row sum) the gradient is
    BIT-equal to autograd's; at D <= 1 024, D % 4 == 0 the row-loop kernel equals the register form bit for bit."""
    from blackjax_amd.targets import ElementwiseRowsTarget
    from blackjax_amd import elementwise as ewm

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    iv = (torch.rand(Dg, device=dev, generator=g) + 0.5)
    mu = torch.randn(Dg, device=dev, generator=g)
    fns = {
        "gaussian": lambda q: -0.5 * (q * q * iv).sum(-1),
        "readme": lambda q: -0.5 * (q * q).sum(-1),
        "shifted_normalised": lambda q: -0.5 * (((q - mu) ** 2) * iv).sum(-1) - 0.5 * Dg * math.log(2 * math.pi),
        "logistic_plus_prior": lambda q: (-torch.nn.functional.softplus(-q * mu)).sum(-1) - 0.5 * (q ** 2).sum(-1) / 4.0,
        "student_t": lambda q: (-2.5 * torch.log1p(q * q / 4.0)).sum(dim=-1),
        "mixed": lambda q: torch.sum(torch.tanh(q) * iv - torch.exp(-q.abs()) + torch.sigmoid(q) / (1.0 + q * q), -1),
    }
    fn = fns[name]
    q = 1.5 * torch.randn(N, Dg, device=dev, generator=g)
    rows = ElementwiseRowsTarget(ewm.trace(fn, Dg, dev), Dg)
    lp, grad = rows(q)
    qa = q.clone().requires_grad_(True)
    lp_a = fn(qa)
    (g_a,) = torch.autograd.grad(lp_a.sum(), qa)
    np.testing.assert_allclose(grad.cpu().numpy(), g_a.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_a.detach().cpu().numpy(), rtol=2e-5, atol=1e-4 * max(1.0, Dg / 1024))
    if name in ("gaussian", "readme"):
        assert torch.equal(grad, g_a)
    if Dg % 4 == 0 and Dg <= 1024:
        lp_r, g_r = bjx.targets.from_elementwise(fn, Dg, device=dev)(q)
        assert torch.equal(lp_r, lp) and torch.equal(g_r, grad)


@pytest.mark.gpu
def test_a_plain_pytorch_function_takes_the_generated_kernel_by_default():
    """`hmc(logdensity_fn)` with a plain PyTorch function (mcmc/hmc.py:90-92: value_and_grad(logdensity_fn)): the first
    call runs under autograd, the function is traced, the generated kernel is checked against that call and serves every
    later one -- same draws as naming `targets.from_elementwise` explicitly; D = 4 096 included.  A function outside the
    element-wise shape, or one declared `no_trace`, stays on autograd (with a RuntimeWarning / silently)."""
    import warnings

    from blackjax_amd import _util

    dev = torch.device("cuda:0")
    for N, Dg, L in ((256, 1024, 6), (64, 4096, 4)):
        sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(Dg) / (Dg - 1))).astype(np.float32), device=dev)
        inv_var, imm = (1.0 / (sig * sig)).contiguous(), (sig * sig).contiguous()
        fn = lambda q: -0.5 * (q * q * inv_var).sum(-1)  # noqa: E731
        fn_eager = bjx.no_trace(lambda q: -0.5 * (q * q * inv_var).sum(-1))
        q0 = sig * torch.randn(N, Dg, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)  # neither path warns
            warnings.filterwarnings("ignore", message=".*not declared recordable.*")
            a_def, a_exp, a_eag = (bjx.hmc(f, 0.2, imm, L) for f in (fn, bjx.targets.from_elementwise(fn, Dg, device=dev), fn_eager))
            s_def, s_exp, s_eag = a_def.init(q0), a_exp.init(q0), a_eag.init(q0)
            for k in bjx.random.split(bjx.random.key(1), 3):
                s_def, i_def = a_def.step(k, s_def)
                s_exp, i_exp = a_exp.step(k, s_exp)
                s_eag, i_eag = a_eag.step(k, s_eag)
        vg = _util.value_and_grad(fn)
        assert [type(v).__name__ for v in vg._bjx_elementwise.values()] == ["DeviceTarget" if Dg <= 1024 else "ElementwiseRowsTarget"]
        assert all(v is None for v in _util.value_and_grad(fn_eager)._bjx_elementwise.values())
        assert torch.equal(s_def.position, s_exp.position) and torch.equal(i_def.is_accepted, i_exp.is_accepted)
        assert torch.equal(s_def.logdensity_grad, s_exp.logdensity_grad)
        # eager autograd: the same gradient bits for this function, logp to rounding -> the same accept decisions
        assert torch.equal(i_def.is_accepted, i_eag.is_accepted)
        np.testing.assert_allclose(s_def.position.cpu().numpy(), s_eag.position.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # outside the shape: autograd, announced once
    outside = lambda q: torch.logsumexp(-0.5 * q * q, -1)  # noqa: E731
    q0 = torch.randn(32, 64, device=dev)
    with pytest.warns(RuntimeWarning, match="evaluated eagerly under torch.autograd"):
        st = bjx.hmc(outside, 0.1, torch.ones(64, device=dev), 3).init(q0)
    qa = q0.clone().requires_grad_(True)
    (g_a,) = torch.autograd.grad(outside(qa).sum(), qa)
    assert torch.equal(st.logdensity_grad, g_a)
    # a traced function whose Python control flow was frozen by the trace fails the first-call check -> autograd
    state = {"sign": -1.0}
    tricky = lambda q: (state["sign"] * 0.5 * q * q).sum(-1) if float(q.abs().max()) < 1e30 else q.sum(-1)  # noqa: E731
    with pytest.warns(RuntimeWarning, match="could not trace|autograd"):
        st = bjx.hmc(tricky, 0.1, torch.ones(64, device=dev), 3).init(q0)
    assert torch.equal(st.logdensity_grad, -q0)
