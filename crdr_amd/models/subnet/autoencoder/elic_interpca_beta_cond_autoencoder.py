"""Synthesis transform of CRDR: ELIC decoder + InterpChAtt (rate) + Fourier-feature beta conditioning (realism)
(src/models/subnet/autoencoder/elic_interpca_beta_cond_autoencoder.py:30-162).  beta -> 2L Fourier features ->
2-layer MLP -> [1, cond_ch, 1, 1]; every bottleneck block projects it to per-channel vectors that are added
after each ReLU (conv epilogue)."""
from __future__ import annotations

from typing import Dict, Union

import torch
import torch.nn as nn

from crdr_amd.models.layer.cheng_nlam import ChengNLAM
from crdr_amd.models.layer.elic_layers import BetaCondResidualBottleneckBlocks, up_conv
from crdr_amd.models.layer.fourier_cond import FourierEmbedding
from crdr_amd.models.layer.hip_layers import HipLinear
from crdr_amd.models.layer.interp_channel_attention import InterpChAtt
from crdr_amd.utils.registry import DECODER_REGISTRY

from .base_autoencoder import BaseDecoder
from .elic_interpca_autoencoder import run_decoder_stages


def weights_init(module):
    """N(0, 0.02) weights / zero bias for every Conv*/Linear (reference :30-40; `weight_init: True` in the configs)."""
    name = module.__class__.__name__
    if name.find("Conv") != -1 or name.find("Linear") != -1:
        if getattr(module, "weight", None) is not None and isinstance(module.weight, nn.Parameter) and module.weight.dim() >= 2:
            module.weight.data.normal_(0.0, 0.02)
            module.bias.data.fill_(0)


class _Mlp(nn.Module):
    def __init__(self, enc_ch, cond_ch):
        super().__init__()
        self.add_module("0", HipLinear(enc_ch, cond_ch))
        self.add_module("2", HipLinear(cond_ch, cond_ch))

    def forward(self, e):
        h = getattr(self, "0")(e, act="relu")
        return getattr(self, "2")(h)


@DECODER_REGISTRY.register()
class ElicInterpCaBetaCondDecoder(BaseDecoder):
    def __init__(self, rate_level: int, L: int = 10, max_beta: float = 5.12, cond_ch: int = 512, use_pi: bool = True,
                 include_x: bool = False, weight_init: bool = False, in_ch: int = 192, out_ch: int = 3, main_ch: int = 192,
                 block_mid_ch: int = 192, num_blocks: int = 3, use_tanh: bool = True, pixel_shuffle: bool = False,
                 res_in_res: bool = False, ca_kwargs: Dict = {}):
        super().__init__()
        if use_tanh:
            raise NotImplementedError("use_tanh is False in every CRDR config")
        self.use_tanh = use_tanh
        self.attn1 = ChengNLAM(in_ch)
        self.conv1 = up_conv(in_ch, main_ch, 5, pixel_shuffle)
        self.block1 = BetaCondResidualBottleneckBlocks(main_ch, block_mid_ch, cond_ch, num_blocks, res_in_res)
        self.conv2 = up_conv(main_ch, main_ch, 5, pixel_shuffle)
        self.attn2 = ChengNLAM(main_ch)
        self.block2 = BetaCondResidualBottleneckBlocks(main_ch, block_mid_ch, cond_ch, num_blocks, res_in_res)
        self.conv3 = up_conv(main_ch, main_ch, 5, pixel_shuffle)
        self.block3 = BetaCondResidualBottleneckBlocks(main_ch, block_mid_ch, cond_ch, num_blocks, res_in_res)
        self.conv4 = up_conv(main_ch, out_ch, 5, pixel_shuffle)
        self.stage_names = ("attn1", "conv1", "block1", "conv2", "attn2", "block2", "conv3", "block3", "conv4")
        chans = [in_ch] * 2 + [main_ch] * 7
        self.layer_in_ch_list = list(zip(self.stage_names, chans))
        self.interp_ca_list = nn.ModuleList(InterpChAtt(c, rate_level, **ca_kwargs) for c in chans)
        self.embed = FourierEmbedding(L=L, max_beta=max_beta, use_pi=use_pi, include_x=include_x)
        enc_ch = 2 * L + 1 if include_x else 2 * L
        self.mlp = _Mlp(enc_ch, cond_ch)
        if weight_init:
            self.apply(weights_init)

    static_embed = None  # optional persistent device buffer [1, 2L, 1, 1] holding the Fourier features of the current beta
                         # (set by the trainer so that a captured HIP graph reads beta from device memory)

    def cond_vector(self, beta: Union[float, torch.Tensor], device) -> torch.Tensor:
        if self.static_embed is not None and self.training:
            e = self.static_embed
        else:
            e = self.embed.embed(beta).to(device).reshape(1, -1, 1, 1)  # [1, 2L] computed on the host
        return self.mlp(e)  # [1, cond_ch, 1, 1]

    def forward(self, x, rate_ind, beta):
        cond = self.cond_vector(beta, x.device)
        return run_decoder_stages(self, x, rate_ind, cond=cond)
