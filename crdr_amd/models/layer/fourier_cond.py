"""Fourier features of beta (Agustsson et al., CVPR 2023 conditioning; src/models/layer/fourier_cond.py:12-37):
[sin(2^k b), cos(2^k b)]_{k<L} with b = 2*beta/max_beta - 1 (optionally times pi). Host-side: 2L numbers."""
from __future__ import annotations

import math
from typing import Union

import torch


class FourierEmbedding:
    def __init__(self, L: int, max_beta: float, use_pi: bool = True, include_x: bool = False) -> None:
        self.L, self.max_beta, self.include_x = L, max_beta, include_x
        self.freq = torch.pow(torch.tensor([2.0]), torch.arange(L))
        if use_pi:
            self.freq = self.freq * math.pi

    def embed(self, beta: Union[int, float, torch.Tensor]) -> torch.Tensor:
        if isinstance(beta, (int, float)):
            beta = torch.tensor([float(beta)])
        beta = beta.detach().float().cpu()
        assert beta.ndim == 1 and 0 <= float(beta.min()) and float(beta.max()) <= self.max_beta
        nb = (beta / self.max_beta - 0.5) * 2
        out = torch.cat([torch.sin(nb * self.freq), torch.cos(nb * self.freq)], dim=0)
        if self.include_x:
            out = torch.cat([nb, out], dim=0)
        return out.unsqueeze(0)
