"""PatchGAN-like discriminator used per rate level in stage 3
(src/models/discriminator/clic21_gvae_discriminator.py:12-50, `norm_type: none`): conv3 s1, [conv3 s2, conv3 s1]x3,
conv3 s2, conv3 head; LeakyReLU(0.2) fused into every conv but the head.  Parameter keys `model.{0,2,...,16}`."""
from __future__ import annotations

import torch.nn as nn

from crdr_amd.hip import chain as CH
from crdr_amd.models.layer.hip_layers import HipConv2d, to_image_nhwc
from crdr_amd.utils.registry import DISCRIMINATOR_REGISTRY

from .base_discriminator import BaseDiscriminator


class _Blocks(nn.Module):
    def __init__(self, in_ch, main_ch, out_ch, kw, num_downscale, head=True):
        super().__init__()
        plan = [(in_ch, main_ch, 1), (main_ch, main_ch, 2)]
        c = main_ch
        for _ in range(num_downscale - 1):
            o = min(c * 2, main_ch * 8)
            plan += [(c, o, 1), (o, o, 2)]
            c = o
        self.n_act = len(plan)
        for i, (ci, co, s) in enumerate(plan):
            self.add_module(str(2 * i), HipConv2d(ci, co, kw, stride=s, padding=kw // 2))
        self.head = head
        if head:
            self.add_module(str(2 * self.n_act), HipConv2d(c, out_ch, 3, stride=1, padding=1))

    def _chain(self) -> "CH.ChainSpec":
        """the conv trunk as one hand-scheduled autograd node: LeakyReLU backward fused into the input-gradient convs,
        bias gradients from their epilogues (crdr_amd/hip/chain.py)"""
        sp = self.__dict__.get("_chain_spec")
        if sp is None:
            layers = [CH.Layer(getattr(self, str(2 * i)), "lrelu") for i in range(self.n_act)]
            if self.head:
                layers.append(CH.Layer(getattr(self, str(2 * self.n_act)), None))
            else:
                layers[-1] = CH.Layer(layers[-1].conv, None)
            sp = CH.ChainSpec([[CH.Unit(layers, residual=False)]], name="disc")
            self.__dict__["_chain_spec"] = sp
        return sp

    def forward(self, x):
        if self.head:
            return CH.run_chain(x, self._chain())[0]
        for i in range(self.n_act):
            x = getattr(self, str(2 * i))(x, act="lrelu")
        return x


@DISCRIMINATOR_REGISTRY.register()
class CLIC21GVAEDiscriminator(BaseDiscriminator):
    def __init__(self, in_ch=3, out_ch=1, main_ch=64, norm_type: str = "BN", num_downscale: int = 4):
        super().__init__()
        if norm_type != "none":
            raise NotImplementedError("CRDR uses norm_type: none (config/crdr_stage_3.yaml:23)")
        self.model = _Blocks(in_ch, main_ch, out_ch, 3, num_downscale)

    def forward(self, input, **kwargs):
        return self.model(to_image_nhwc(input))
