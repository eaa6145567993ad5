"""Auxiliary (quantile) loss of the factorised prior: sum |logits_cdf(quantiles) - target| with the density
network detached -- 3 values per channel.  Runs through the same HIP kernel as the likelihood by evaluating the
bottleneck's logits at the quantiles."""
from __future__ import annotations

import torch

from crdr_amd.hip import aux_ops


def eb_aux_loss(eb) -> torch.Tensor:
    return aux_ops.eb_quantile_loss(eb.quantiles, eb.packed_params().detach(), eb.target)
