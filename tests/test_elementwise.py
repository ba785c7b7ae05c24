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
    with pytest.raises(NotImplementedError, match="1 024"):
        bjx.targets.from_elementwise(FUNCTIONS["readme"], 2048, device="cpu")


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
