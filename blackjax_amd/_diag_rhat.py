"""``potential_scale_reduction`` on device (blackjax/diagnostics.py:39-89)."""
from __future__ import annotations

import torch


def potential_scale_reduction(input_array: torch.Tensor, chain_axis: int = 0,
                              sample_axis: int = 1) -> torch.Tensor:
    """Gelman-Rubin R-hat with the chain and sample axes squeezed (diagnostics.py:39-89)."""
    assert input_array.shape[chain_axis] > 1, \
        "potential_scale_reduction as implemented only works for two or more chains."
    x = torch.movedim(input_array, (chain_axis, sample_axis), (0, 1))
    if x.dtype not in (torch.float32, torch.float64):
        x = x.double()
    num_samples = x.shape[1]
    per_chain_mean = x.mean(dim=1)
    per_chain_var = x.var(dim=1, unbiased=True)
    between = num_samples * per_chain_mean.var(dim=0, unbiased=True)
    within = per_chain_var.mean(dim=0)
    return torch.sqrt((between / within + num_samples - 1) / num_samples)
