"""Non-saturating GAN loss: BCE-with-logits against a constant label, mean over logits; the generator-side value
carries the loss weight (src/losses/gan_loss.py:10-31).  `forward_diff(p, q, ...)` evaluates the loss of the
logit difference p - q (the relativistic form used by the stage-3 trainer) in one fused reduction."""
from __future__ import annotations

import torch
import torch.nn as nn

from crdr_amd.hip import functional as HF
from crdr_amd.utils.registry import LOSS_REGISTRY


@LOSS_REGISTRY.register()
class VanillaGANLoss(nn.Module):
    def __init__(self, loss_weight: float, real_label: float = 1.0, fake_label: float = 0.0, loss_reduction: str = "mean"):
        super().__init__()
        assert loss_reduction == "mean"
        self.lamb_gan = loss_weight
        self.real_label_val, self.fake_label_val = real_label, fake_label

    def forward_diff(self, p, q, is_real: bool, is_disc: bool = False):
        target = self.real_label_val if is_real else self.fake_label_val
        loss = (HF.bce_diff_sum(p, q, target) / p.numel()).reshape(())
        return loss if is_disc else self.lamb_gan * loss

    def forward(self, x, is_real: bool, is_disc: bool = False, **kwargs):
        return self.forward_diff(x, torch.zeros_like(x), is_real, is_disc)
