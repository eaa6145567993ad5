"""Additive uniform noise for the rate estimate, straight-through rounding for the decoder input
(src/models/subnet/entropy_model/ste_gaussian_conditional.py:10-27)."""
from __future__ import annotations

import torch
from torch import Tensor

from crdr_amd.hip import functional as HF
from crdr_amd.utils.registry import ENTROPYMODEL_REGISTRY

from .gaussian_conditional import GaussianMeanScaleConditional


@ENTROPYMODEL_REGISTRY.register()
class SteGaussianMeanScaleConditional(GaussianMeanScaleConditional):
    def __init__(self, scale_bound=None, entropy_quant_type: str = "noise", **kwargs) -> None:
        super().__init__(scale_bound=scale_bound)
        assert entropy_quant_type == "noise", "only noise quantisation is supported for the entropy estimate"
        self.entropy_quant_type = entropy_quant_type

    def forward(self, y: Tensor, params: Tensor, is_train: bool = True, noise: Tensor = None, want_bits: bool = False):
        mean, std = params.chunk(2, 1)
        return self.forward_split(y, mean, std, is_train=is_train, noise=noise, want_bits=want_bits)

    def forward_split(self, y, mean, std, is_train=True, noise=None, want_bits=False, want_lik=True):
        """One fused launch: y_hat (STE-rounded), noisy + quantised likelihoods and both per-image bit sums.
        -> (y_hat, likelihood[, bits_noisy, bits_quant])"""
        if is_train and noise is None:
            noise = torch.rand(y.shape, device=y.device).contiguous(memory_format=torch.channels_last) - 0.5
        yhat, bn, bq, ln, lq = HF.gauss_cond(y, mean, std, noise if is_train else None, self.scale_bound,
                                             self.likelihood_bound, want_lik)
        lik = ln if is_train else lq
        if want_bits:
            return yhat, lik, (bn if is_train else bq), bq, lq
        return yhat, lik
