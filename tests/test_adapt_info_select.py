"""Which per-step record ``window_adaptation`` keeps (host logic, no GPU): only the DEFAULT ``adaptation_info_fn`` is
downgraded to per-chain scalars above ``ALL_INFO_MAX_BYTES``; an explicit ``return_all_adapt_info`` (the reference's
adaptation/base.py:32-36) or a user function is honoured as given."""
import inspect
import warnings

import pytest
import torch

from blackjax_amd import adaptation as bad


def test_default_is_a_private_sentinel_that_records_everything():
    for fn in (bad.window_adaptation, bad.staged_adaptation):
        assert inspect.signature(fn).parameters["adaptation_info_fn"].default is bad._default_adapt_info
    assert bad._default_adapt_info is not bad.return_all_adapt_info
    assert bad._default_adapt_info(1, 2, 3) == bad.return_all_adapt_info(1, 2, 3) == bad.AdaptationInfo(1, 2, 3)


def test_only_the_default_is_downgraded(monkeypatch):
    monkeypatch.setattr(bad, "ALL_INFO_MAX_BYTES", 11 * 4 * 64 * 64)
    bad._WARNED_INFO.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert bad._select_info_fn(bad._default_adapt_info, 64, 64, True) == (bad._default_adapt_info, False)  # at the bound
        assert bad._select_info_fn(bad.return_all_adapt_info, 4096, 4096, True) == (bad.return_all_adapt_info, False)
        user = lambda s, i, a: None  # noqa: E731
        assert bad._select_info_fn(user, 4096, 4096, True) == (user, False)
        assert bad._select_info_fn(None, 4096, 4096, True) == (None, True)
        assert bad._select_info_fn(None, 4096, 4096, False) == (None, False)  # dense Welford is never in place
        filt = bad.get_filter_adapt_info_fn(info_keys={"acceptance_rate"})
        assert bad._select_info_fn(filt, 4096, 4096, True) == (filt, True)
        keeps = bad.get_filter_adapt_info_fn(adapt_state_keys={"imm_state"})
        assert bad._select_info_fn(keeps, 4096, 4096, True) == (keeps, False)
    with pytest.warns(RuntimeWarning, match="per-chain scalars"):
        assert bad._select_info_fn(bad._default_adapt_info, 64, 65, True) == (bad._scalars_only_adapt_info, True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the same shape again: announced once
        assert bad._select_info_fn(bad._default_adapt_info, 64, 65, True)[0] is bad._scalars_only_adapt_info
    with pytest.warns(RuntimeWarning):  # another shape: announced again
        bad._select_info_fn(bad._default_adapt_info, 128, 65, True)


def test_scalars_only_record_drops_the_metric_by_name_when_n_equals_d():
    from blackjax_amd.hmc import HMCInfo, HMCState

    n = d = 8
    state = HMCState(torch.zeros(n, d), torch.zeros(n), torch.zeros(n, d))
    info = HMCInfo(torch.zeros(n, d), torch.ones(n), torch.ones(n, dtype=torch.bool), torch.zeros(n, dtype=torch.bool),
                   torch.zeros(n), None, 3)
    ss = bad.DualAveragingAdaptationState(torch.zeros(n), torch.zeros(n), 1, torch.zeros(n), torch.zeros(n))
    for imm in (torch.ones(d), torch.ones(n, d)):  # the shared (D,) diagonal looks like a per-chain scalar when N == D
        ws = bad.StagedAdaptationState(ss, bad.MassMatrixAdaptationState(imm, bad.WelfordAlgorithmState(
            torch.zeros(n, d), torch.zeros(n, d), 0)), torch.ones(n), imm)
        rec = bad._scalars_only_adapt_info(state, info, ws)
        assert rec.state.position is None and rec.state.logdensity is not None
        assert rec.info.momentum is None and rec.info.acceptance_rate is not None and rec.info.num_integration_steps == 3
        assert rec.adaptation_state.inverse_mass_matrix is None
        assert rec.adaptation_state.imm_state.inverse_mass_matrix is None
        assert rec.adaptation_state.imm_state.wc_state.mean is None
        assert rec.adaptation_state.step_size is not None and rec.adaptation_state.ss_state.log_step_size is not None
