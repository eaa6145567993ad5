"""Hyper analysis / synthesis of Minnen & Singh (ICIP 2020) as used by CRDR
(src/models/subnet/hyperprior/minnen20_hyperprior.py:9-58): z = conv5s2(relu(conv5s2(relu(conv3 y)))); the
decoder has two independent branches (mean, scale), each ConvT5s2 -> ReLU -> ConvT5s2 -> ReLU -> ConvT3s1."""
from __future__ import annotations

import torch
import torch.nn as nn

from crdr_amd.models.layer.hip_layers import HipConv2d, HipConvTranspose2d
from crdr_amd.utils.registry import HYPERDECODER_REGISTRY, HYPERENCODER_REGISTRY


@HYPERENCODER_REGISTRY.register()
class Minnen20HyperEncoder(nn.Module):
    def __init__(self, bottleneck_y: int = 320, bottleneck_z: int = 192):
        super().__init__()
        self.conv1 = HipConv2d(bottleneck_y, 320, 3, stride=1, padding=1)
        self.conv2 = HipConv2d(320, 256, 5, stride=2, padding=2)
        self.conv3 = HipConv2d(256, bottleneck_z, 5, stride=2, padding=2)
        self.num_downscale = 2
        self.n_downsampling_layers = 2
        self.latent_ch = bottleneck_z

    def forward(self, x):
        x = self.conv1(x, act="relu")
        x = self.conv2(x, act="relu")
        return self.conv3(x)


class HyperDecoderBlock(nn.Module):
    def __init__(self, in_ch: int = 192, out_ch: int = 320):
        super().__init__()
        self.conv1 = HipConvTranspose2d(in_ch, 192, 5, stride=2, padding=2, output_padding=1)
        self.conv2 = HipConvTranspose2d(192, 256, 5, stride=2, padding=2, output_padding=1)
        self.conv3 = HipConvTranspose2d(256, out_ch, 3, stride=1, padding=1)

    def forward(self, x):
        x = self.conv1(x, act="relu")
        x = self.conv2(x, act="relu")
        return self.conv3(x)


@HYPERDECODER_REGISTRY.register()
class Minnen20HyperDecoder(nn.Module):
    def __init__(self, bottleneck_z: int = 192, hyper_out_ch: int = 640):
        super().__init__()
        assert hyper_out_ch % 2 == 0
        self.hd_mu = HyperDecoderBlock(bottleneck_z, hyper_out_ch // 2)
        self.hd_std = HyperDecoderBlock(bottleneck_z, hyper_out_ch // 2)

    def forward_pair(self, x):
        return self.hd_mu(x), self.hd_std(x)

    def forward(self, x):
        return torch.cat(self.forward_pair(x), dim=1)
