"""potential_scale_reduction (blackjax/diagnostics.py:39-89): device implementation vs a NumPy
restatement of the reference formula, and the reference test's property (R-hat ~ 1 for iid normals,
tests/test_diagnostics.py:42-89, rtol 1e-3)."""
import numpy as np
import torch

from blackjax_amd.diagnostics import potential_scale_reduction, rhat


def _np_rhat(x):
    n = x.shape[1]
    m = x.mean(1, keepdims=True)
    v = x.var(1, ddof=1, keepdims=True)
    b = n * m.var(0, ddof=1, keepdims=True)
    w = v.mean(0, keepdims=True)
    return np.sqrt((b / w + n - 1) / n).squeeze()


def test_rhat_matches_reference_formula_and_iid_property():
    rng = np.random.default_rng(32)
    for shape in [(), (3,), (5, 7)]:
        x = rng.standard_normal((10, 5000) + shape)
        r = potential_scale_reduction(torch.as_tensor(x)).numpy()
        np.testing.assert_allclose(r, _np_rhat(x), rtol=1e-10)
        np.testing.assert_allclose(r, 1.0, rtol=1e-3)
    # un-mixed chains are flagged
    y = rng.standard_normal((4, 500)) + np.arange(4)[:, None] * 3
    assert float(rhat(torch.as_tensor(y))) > 1.5
    # axes arguments
    z = rng.standard_normal((200, 6, 3))
    np.testing.assert_allclose(potential_scale_reduction(torch.as_tensor(z), chain_axis=1, sample_axis=0).numpy(),
                               _np_rhat(np.moveaxis(z, (1, 0), (0, 1))), rtol=1e-10)
