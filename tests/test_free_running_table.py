"""Host table of the free-running warm-up (blackjax_amd.adaptation.free_running_table): schedule flags
and per-step scalars against the Stan window schedule (staged_adaptation.py:315-405) and the formulas
of dual_averaging.py:117-122 / mass_matrix.py:339-343.  CPU only."""
import numpy as np
import pytest

from blackjax_amd import _lib
from blackjax_amd.adaptation import build_schedule, free_running_table

AT = _lib.NUTS_AT


@pytest.mark.parametrize("T,shrink", [(1000, 0.0), (150, 2.5), (19, 0.0), (60, 0.0)])
def test_table_follows_schedule(T, shrink):
    tab = free_running_table(T, shrink)
    sched = build_schedule(T)
    assert tab.shape == (T, _lib.NUTS_ADAPT_COLS) and tab.dtype == np.float32
    da_step, wel = 1, 0
    for t, (stage, end) in enumerate(sched):
        assert int(tab[t, AT["FLAGS"]]) == (1 if stage == 1 else 0) | (2 if end else 0)
        if stage == 1:
            wel += 1
            assert tab[t, AT["WEL_N"]] == wel
        else:
            assert tab[t, AT["WEL_N"]] == 0
        assert tab[t, AT["DA_REG"]] == np.float32(da_step + 10.0)
        assert tab[t, AT["DA_INV_REG"]] == np.float32(1.0) / np.float32(da_step + 10.0)
        np.testing.assert_allclose(tab[t, AT["DA_ETA"]], da_step ** -0.75, rtol=1e-7)
        np.testing.assert_allclose(tab[t, AT["DA_COEF"]], np.sqrt(da_step) / 0.05, rtol=2e-7)
        da_step += 1
        if end:
            denom = np.float32(wel + 5) + np.float32(shrink)
            assert tab[t, AT["FIN_NM1"]] == wel - 1
            assert tab[t, AT["FIN_BETA_DATA"]] == np.float32(wel) / denom
            assert tab[t, AT["FIN_BETA_PREV"]] == np.float32(shrink) / denom
            assert tab[t, AT["FIN_REG"]] == (np.float32(5.0) / denom) * np.float32(1e-3)
            wel, da_step = 0, 1
        else:
            assert not tab[t, AT["FIN_NM1"]:AT["FIN_REG"] + 1].any()
    if T >= 20:
        assert any(end for _, end in sched), "a schedule of >= 20 steps has at least one window end"


def test_staged_adaptation_entry_point_validation_and_schedule_fn():
    """blackjax/adaptation/staged_adaptation.py:519-983: the engine entry point -- recipe names, the
    reference's n_chains error, out-of-scope metrics, and a custom schedule_fn's shape check (no GPU
    needed: construction-time behaviour only)."""
    import pytest

    import blackjax_amd as bjx
    from blackjax_amd import adaptation as bad

    fn = lambda q: q.sum(-1)
    for name in ("welford_diag", "welford_dense"):
        assert callable(bjx.staged_adaptation(bjx.hmc, fn, metric=name, num_integration_steps=3).run)
    with pytest.raises(ValueError, match="n_chains > 1 is only supported"):
        bjx.staged_adaptation(bjx.hmc, fn, n_chains=4)
    with pytest.raises(ValueError, match="n_chains must be >= 1"):
        bjx.staged_adaptation(bjx.hmc, fn, n_chains=0)
    for bad_metric in ("auto", "fisher_diag"):
        with pytest.raises(NotImplementedError):
            bjx.staged_adaptation(bjx.hmc, fn, metric=bad_metric)
    with pytest.raises(TypeError, match="metric must be a str"):  # the reference's own error (staged_adaptation.py:509-515)
        bjx.staged_adaptation(bjx.hmc, fn, metric=object())
    assert bad._as_schedule([[0, False], [1, False], [1, True]], 3) == [(0, False), (1, False), (1, True)]
    with pytest.raises(ValueError):
        bad._as_schedule([[0, False]], 3)
    with pytest.raises(ValueError):
        bad._as_schedule([[2, False]], 1)
