"""Conv2d / ConvTranspose2d / Linear whose arithmetic is the fused implicit-GEMM HIP kernel.

They subclass the torch modules only for parameter creation, default initialisation and state-dict layout
(OIHW / IOHW keys `weight`, `bias` -- the reference's checkpoint schema); `forward` never calls ATen compute.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from crdr_amd.hip import functional as HF


class HipConv2d(nn.Conv2d):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1, padding: int = 0):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=padding)
        self.spec = HF.ConvSpec(in_ch, out_ch, kernel_size, stride, padding, transposed=False)

    def forward(self, x, *, act: Optional[str] = None, vec2=None, res=None, affine=None, gate=None):
        return HF.fused_conv(x, self.weight, self.bias, self.spec, act=act, vec2=vec2, res=res, affine=affine, gate=gate)


class HipSpectralNormConv2d(nn.Module):
    """`torch.nn.utils.spectral_norm(nn.Conv2d(...))` (the reference's hific_discriminator.py:10-13) with the same
    state-dict entries -- `weight_orig`, `weight_u`, `weight_v`, `bias` -- and the same behaviour: one power iteration per
    training-mode forward (u, v updated in place), none in eval; the conv consumes weight_orig / sigma."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1, padding: int = 0, eps: float = 1e-12):
        super().__init__()
        ref = nn.Conv2d(in_ch, out_ch, kernel_size, stride=stride, padding=padding)  # default init of the wrapped conv
        self.weight_orig = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        k = in_ch * kernel_size * kernel_size
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(out_ch), dim=0, eps=eps))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn(k), dim=0, eps=eps))
        self.register_buffer("_w_sn", torch.zeros(out_ch, in_ch, kernel_size, kernel_size), persistent=False)
        self.eps = eps
        self.spec = HF.ConvSpec(in_ch, out_ch, kernel_size, stride, padding, transposed=False)

    def forward(self, x, *, act: Optional[str] = None):
        w = HF.spectral_norm_weight(self.weight_orig, self.weight_u, self.weight_v, self.training, self._w_sn, self.eps)
        self.spec.mark_stale()  # _w_sn was rewritten in place
        return HF.fused_conv(x, w, self.bias, self.spec, act=act, return_wgrad=True)


class HipConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1, padding: int = 0, output_padding: int = 0):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=padding, output_padding=output_padding)
        self.spec = HF.ConvSpec(in_ch, out_ch, kernel_size, stride, padding, transposed=True, out_pad=output_padding)

    def forward(self, x, *, act: Optional[str] = None, affine=None):
        return HF.fused_conv(x, self.weight, self.bias, self.spec, act=act, affine=affine)


class HipLinear(nn.Linear):
    """y = x W^T + b on a [1, in] row vector, run as a 1x1 conv over a single pixel."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__(in_features, out_features)
        self.spec = HF.ConvSpec(in_features, out_features, 1, 1, 0)

    def forward(self, x, *, act: Optional[str] = None):
        x4 = x.reshape(x.shape[0], x.shape[1], 1, 1) if x.dim() == 2 else x
        return HF.fused_conv(x4, self.weight, self.bias, self.spec, act=act)


def to_image_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,3,H,W] any layout -> view with NHWC memory padded to 4 channels (4th lane zero)."""
    from crdr_amd.hip import ops
    y, _ = ops.nhwc(x)
    return y
