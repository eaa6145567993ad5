"""Interpolation channel attention (Sun et al., ACM MM 2022) as used by CRDR for rate control
(src/models/layer/interp_channel_attention.py:16-73): per-channel `softplus(lerp(W))*x + lerp(b)` where the
raw weights of the two neighbouring rate levels are interpolated BEFORE the softplus.

Here the module only produces the per-channel (scale, shift) vectors (one tiny kernel); applying them is the
epilogue of the neighbouring conv, or `forward(x, q)` for stand-alone use."""
from __future__ import annotations

import math
from typing import Tuple, Union

import torch
import torch.nn as nn

from crdr_amd.hip import functional as HF


class InterpChAtt(nn.Module):
    def __init__(self, ch: int, rate_level: int, actv: str = "identity", use_interp: bool = False, use_bias: bool = False) -> None:
        super().__init__()
        if actv != "softplus" or not use_interp:
            raise NotImplementedError("CRDR uses actv=softplus, use_interp=True (config/_base_/model/*interp_ca*.yaml)")
        w = torch.ones(rate_level, 1, ch, 1, 1, dtype=torch.float32) * math.log(math.e - 1)  # softplus(w) == 1
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(rate_level, 1, ch, 1, 1, dtype=torch.float32)) if use_bias else None
        self.use_interp, self.use_bias, self.rate_level = use_interp, use_bias, rate_level

    @staticmethod
    def _q(rate_ind: Union[float, torch.Tensor]) -> float:
        if isinstance(rate_ind, torch.Tensor):
            assert rate_ind.numel() == 1, "one rate index per batch (batch_rate_ind_sample is unsupported upstream too)"
            rate_ind = float(rate_ind.item())
        return float(rate_ind)

    def vectors(self, rate_ind) -> Tuple[torch.Tensor, torch.Tensor]:
        q = self._q(rate_ind)
        assert 0 <= q <= self.rate_level - 1, f"rate_ind = {q} should be in [0, {self.rate_level - 1}]"
        return HF.interp_ca_vectors(self.weight, self.bias, q)

    def forward(self, x, rate_ind):
        s, t = self.vectors(rate_ind)
        return HF.affine(x, s, t)
