"""LPIPS (AlexNet) perceptual loss on the HIP kernels (src/losses/perceptual_loss.py:11-30 wraps lpips==0.1.4,
which is third-party and not installed here): scaling layer -> AlexNet conv1..5 (ReLU fused, max-pool 3/2 before
conv2 and conv3) -> per layer: unit-normalise over channels, squared difference, 1x1 `lin` weights, spatial
mean; summed over the five layers.  The network is frozen; gradients flow to the reconstructed image only.

Weights: torchvision's AlexNet + lpips' linear heads are not available offline -- `load_lpips_weights` accepts
the two upstream state dicts when supplied; otherwise the module keeps its (seedable) random init, which is what
the throughput benchmark uses (stated in bench.py's output)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from crdr_amd.hip import functional as HF
from crdr_amd.hip import ops
from crdr_amd.models.layer.hip_layers import HipConv2d
from crdr_amd.utils.registry import LOSS_REGISTRY

ALEX_CFG = ((3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1))
LPIPS_SHIFT = (-0.030, -0.088, -0.188)
LPIPS_SCALE = (0.458, 0.448, 0.450)
_TV_IDX = (0, 3, 6, 8, 10)  # torchvision alexnet.features indices of the convs


class LpipsAlex(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.ModuleList(HipConv2d(ci, co, k, stride=s, padding=p) for ci, co, k, s, p in ALEX_CFG)
        self.lin = nn.ParameterList(nn.Parameter(torch.rand(co) * 0.1) for _, co, _, _, _ in ALEX_CFG)
        self.register_buffer("in_scale", 1.0 / torch.tensor(LPIPS_SCALE))
        self.register_buffer("in_shift", -torch.tensor(LPIPS_SHIFT) / torch.tensor(LPIPS_SCALE))
        for p in self.parameters():
            p.requires_grad_(False)

    def load_lpips_weights(self, alexnet_features_sd: Dict[str, torch.Tensor], lpips_lin_sd: Dict[str, torch.Tensor]) -> None:
        for i, j in enumerate(_TV_IDX):
            self.net[i].weight.data.copy_(alexnet_features_sd[f"{j}.weight"])
            self.net[i].bias.data.copy_(alexnet_features_sd[f"{j}.bias"])
            self.lin[i].data.copy_(lpips_lin_sd[f"lin{i}.model.1.weight"].reshape(-1))

    def features(self, x):
        x, _ = ops.nhwc(x)
        x = HF.affine(x, self.in_scale, self.in_shift)
        feats = []
        for i, conv in enumerate(self.net):
            if i in (1, 2):
                x = HF.maxpool3s2(x)
            x = conv(x, act="relu")
            feats.append(x)
        return feats

    def forward(self, real, fake):
        with torch.no_grad():
            f_real = self.features(real.detach())
        f_fake = self.features(fake)
        total = None
        for a, b, lin in zip(f_real, f_fake, self.lin):
            d = HF.lpips_layer(a, b, lin)
            total = d if total is None else total + d
        return total  # [N]


@LOSS_REGISTRY.register()
class LPIPSLoss(nn.Module):
    def __init__(self, loss_weight: float, range_norm: bool = False, net: str = "alex"):
        super().__init__()
        assert net == "alex" and not range_norm
        self.lamb_lpips = loss_weight
        self.lpips = LpipsAlex()

    def forward(self, real_images, fake_images):
        return self.lamb_lpips * self.lpips(real_images, fake_images).mean()
