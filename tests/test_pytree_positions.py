"""blackjax_amd.util.ravel_chain_pytree / flat_logdensity: pytree positions (dict of parameters) for an
engine whose positions are one (N, D) tensor.  Host logic, CPU only."""
import collections

import pytest
import torch

from blackjax_amd.util import flat_logdensity, ravel_chain_pytree


def test_roundtrip_order_and_shapes():
    N = 5
    tree = {"scale": torch.arange(N * 3.0).reshape(N, 3), "loc": torch.arange(N * 2.0).reshape(N, 2) + 100,
            "nested": [torch.full((N,), 7.0), (torch.ones(N, 2, 2),)]}
    flat, unravel = ravel_chain_pytree(tree)
    assert flat.shape == (N, 2 + 1 + 4 + 3) and flat.dtype == torch.float32
    # jax.tree_util order: dict keys sorted -> loc, nested[0], nested[1][0], scale
    assert torch.equal(flat[:, :2], tree["loc"]) and torch.equal(flat[:, 2], tree["nested"][0])
    assert torch.equal(flat[:, 7:], tree["scale"])
    back = unravel(flat)
    assert list(back) == list(tree)  # insertion order of the user's dict is kept
    assert torch.equal(back["scale"], tree["scale"]) and back["nested"][1][0].shape == (N, 2, 2)
    assert isinstance(back["nested"], list) and isinstance(back["nested"][1], tuple)
    # any number of rows (compacted batches of the NUTS kernels)
    assert unravel(flat[:2])["loc"].shape == (2, 2)


def test_namedtuple_and_errors():
    P = collections.namedtuple("P", ["a", "b"])
    tree = P(torch.zeros(4, 2), torch.ones(4))
    flat, unravel = ravel_chain_pytree(tree)
    back = unravel(flat)
    assert isinstance(back, P) and torch.equal(back.b, tree.b)
    with pytest.raises(ValueError):
        ravel_chain_pytree({"a": torch.zeros(4, 2), "b": torch.zeros(3)})
    with pytest.raises(ValueError):
        unravel(torch.zeros(4, 7))


def test_flat_logdensity_gradient_matches_tree_gradient():
    N = 6
    g = torch.Generator().manual_seed(0)
    tree = {"loc": torch.randn(N, 3, generator=g), "log_scale": torch.randn(N, generator=g)}

    def logdensity(p):  # batched over chains, written against the pytree
        return -0.5 * (p["loc"] ** 2).sum(-1) * torch.exp(-2 * p["log_scale"]) - 3 * p["log_scale"]

    flat, unravel = ravel_chain_pytree(tree)
    fn = flat_logdensity(logdensity, unravel)
    q = flat.clone().requires_grad_(True)
    lp = fn(q)
    (gq,) = torch.autograd.grad(lp.sum(), q)
    leaves = {k: v.clone().requires_grad_(True) for k, v in tree.items()}
    g_loc, g_ls = torch.autograd.grad(logdensity(leaves).sum(), [leaves["loc"], leaves["log_scale"]])
    assert torch.allclose(lp, logdensity(tree))
    assert torch.allclose(gq[:, :3], g_loc) and torch.allclose(gq[:, 3], g_ls)
