"""Balle et al. (ICLR 2018) transforms: four stride-2 5x5 (transposed) convs with GDN / inverse GDN in between
(src/models/subnet/autoencoder/balle18_autoencoder.py:10-58; an ablation baseline of the reference, no shipped CRDR config
uses it).  Registered so that a config naming `Balle18Encoder` / `Balle18Decoder` builds; same state-dict keys
(`conv.{0,2,4,6}.{weight,bias}`, `conv.{1,3,5}.{beta,gamma,...}`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from crdr_amd.models.layer.gdn import GDN
from crdr_amd.models.layer.hip_layers import HipConv2d, HipConvTranspose2d, to_image_nhwc
from crdr_amd.utils.registry import DECODER_REGISTRY, ENCODER_REGISTRY

from .base_autoencoder import BaseDecoder, BaseEncoder


class _Chain(nn.Module):
    def __init__(self, layers):
        super().__init__()
        for i, l in enumerate(layers):
            self.add_module(str(i), l)
        self.n = len(layers)

    def forward(self, x):
        for i in range(self.n):
            x = getattr(self, str(i))(x)
        return x


@ENCODER_REGISTRY.register()
class Balle18Encoder(BaseEncoder):
    def __init__(self, in_ch: int = 3, out_ch: int = 192, main_ch: int = 192):
        super().__init__()
        self.conv = _Chain([HipConv2d(in_ch, main_ch, 5, stride=2, padding=2), GDN(main_ch),
                            HipConv2d(main_ch, main_ch, 5, stride=2, padding=2), GDN(main_ch),
                            HipConv2d(main_ch, main_ch, 5, stride=2, padding=2), GDN(main_ch),
                            HipConv2d(main_ch, out_ch, 5, stride=2, padding=2)])
        self.num_downscale = 4
        self.latent_ch = out_ch

    def forward(self, x):
        return self.conv(to_image_nhwc(x))


@DECODER_REGISTRY.register()
class Balle18Decoder(BaseDecoder):
    def __init__(self, in_ch: int = 192, out_ch: int = 3, main_ch: int = 192, use_tanh: bool = True):
        super().__init__()
        up = lambda a, b: HipConvTranspose2d(a, b, 5, stride=2, padding=2, output_padding=1)
        self.conv = _Chain([up(in_ch, main_ch), GDN(main_ch, inverse=True), up(main_ch, main_ch), GDN(main_ch, inverse=True),
                            up(main_ch, main_ch), GDN(main_ch, inverse=True), up(main_ch, out_ch)])
        self.use_tanh = use_tanh

    def forward(self, x):
        x = self.conv(x)
        return torch.tanh(x) if self.use_tanh else x
