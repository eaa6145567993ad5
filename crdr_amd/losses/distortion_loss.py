"""MSE on [0,1]-scaled images times a weight (src/losses/distortion_loss.py:12-46): one fused
sum-of-squared-differences reduction (and its backward) instead of normalise / subtract / square / mean."""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor

from crdr_amd.hip import functional as HF
from crdr_amd.utils.registry import LOSS_REGISTRY


@LOSS_REGISTRY.register()
class MSELoss(nn.Module):
    def __init__(self, loss_weight: float, normalize_img: bool = True, mse_scale: str = "0_1"):
        super().__init__()
        assert normalize_img
        assert mse_scale in ("0_255", "0_1"), f'mse_scale should be "0_255" or "0_1", but {mse_scale}'
        self.lamb_mse = loss_weight
        self.range_scale = 0.5 if mse_scale == "0_1" else 127.5  # (x+1)/2 [*255] is affine: differences scale by this

    def forward(self, real_images: Tensor, fake_images: Tensor, **kwargs):
        n = real_images.numel()
        return (HF.sqdiff_sum(real_images, fake_images) * (self.lamb_mse * self.range_scale ** 2 / n)).reshape(())
