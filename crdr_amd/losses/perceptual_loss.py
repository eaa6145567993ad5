"""LPIPS (AlexNet) perceptual loss on the HIP kernels (src/losses/perceptual_loss.py:11-30 wraps lpips==0.1.4,
which is third-party and not installed here): scaling layer -> AlexNet conv1..5 (ReLU fused, max-pool 3/2 before
conv2 and conv3) -> per layer: unit-normalise over channels, squared difference, 1x1 `lin` weights, spatial
mean; summed over the five layers.  The network is frozen; gradients flow to the reconstructed image only.

Weights: torchvision's AlexNet + lpips' linear heads cannot be downloaded here, so they are supplied as ONE file,
`torch.save({"alexnet_features": torchvision.models.alexnet(weights=...).features.state_dict(),
             "lpips_lin": {k: v for k, v in lpips.LPIPS(net="alex").state_dict().items() if k.startswith("lin")}}, path)`,
named by the loss option `weights:` (YAML / CLI overlay `loss.perceptual_loss.weights=...`) or the environment variable
CRDR_LPIPS_WEIGHTS.  Without it LPIPSLoss refuses to be built -- a random-feature distance is not LPIPS -- unless
`allow_random_weights: true` / CRDR_ALLOW_RANDOM_LPIPS=1 is set explicitly (throughput benchmark, tests)."""
from __future__ import annotations

import logging
import os
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

    def load_lpips_file(self, path: str) -> None:
        """One file holding {"alexnet_features": state dict of torchvision alexnet.features,
        "lpips_lin": lpips.LPIPS(net='alex') state dict (its lin{i}.model.1.weight entries)}."""
        blob = torch.load(path, map_location="cpu")
        if not (isinstance(blob, dict) and "alexnet_features" in blob and "lpips_lin" in blob):
            raise ValueError(f"{path}: expected a dict with the keys 'alexnet_features' and 'lpips_lin' (see crdr_amd/losses/"
                             "perceptual_loss.py)")
        self.load_lpips_weights(blob["alexnet_features"], blob["lpips_lin"])

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
    def __init__(self, loss_weight: float, range_norm: bool = False, net: str = "alex", weights: Optional[str] = None,
                 allow_random_weights: bool = False):
        super().__init__()
        assert net == "alex" and not range_norm
        self.lamb_lpips = loss_weight
        self.lpips = LpipsAlex()
        weights = weights or os.environ.get("CRDR_LPIPS_WEIGHTS")
        if weights:
            self.lpips.load_lpips_file(weights)
            self.pretrained = True
        elif allow_random_weights or os.environ.get("CRDR_ALLOW_RANDOM_LPIPS") == "1":
            logging.getLogger("crdr").warning(
                "LPIPSLoss: NO pretrained weights given -- the perceptual term is a random-feature distance, not LPIPS "
                "(fine for throughput measurements and plumbing tests, wrong for training)")
            self.pretrained = False
        else:
            raise RuntimeError(
                "LPIPSLoss needs the pretrained AlexNet + LPIPS linear-head weights (the reference gets them from the `lpips` "
                "package, src/losses/perceptual_loss.py:23): pass `weights: <file>` in loss.perceptual_loss or set "
                "CRDR_LPIPS_WEIGHTS (file layout: crdr_amd/losses/perceptual_loss.py); to run on random features on purpose "
                "set allow_random_weights: true or CRDR_ALLOW_RANDOM_LPIPS=1")

    def forward(self, real_images, fake_images):
        return self.lamb_lpips * self.lpips(real_images, fake_images).mean()
