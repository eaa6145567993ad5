"""ELIC building blocks (He et al., CVPR 2022) on the fused HIP conv: residual bottleneck `x + 1x1(relu(3x3(
relu(1x1 x))))` with the bias/ReLU/residual (and, for the last block of a stage, the following InterpChAtt
scale+shift) folded into the conv epilogues.  Mirrors src/models/layer/elic_layers.py:14-53 and the beta-cond
variant of src/models/subnet/autoencoder/elic_interpca_beta_cond_autoencoder.py:42-84 (same parameter names)."""
from __future__ import annotations

import torch.nn as nn

from crdr_amd.hip import chain as CH
from crdr_amd.hip import functional as HF

from .hip_layers import HipConv2d, HipConvTranspose2d


def up_conv(in_ch: int, out_ch: int, kernel_size: int, pixel_shuffle: bool):
    assert kernel_size == 5, "only kernel_size=5 (the ELIC setting) is supported"
    if pixel_shuffle:
        raise NotImplementedError("pixel_shuffle up-sampling is not used by any CRDR config")
    return HipConvTranspose2d(in_ch, out_ch, kernel_size, stride=2, padding=2, output_padding=1)


class _ConvSeq(nn.Module):
    """Holds convs under the reference's nn.Sequential indices 0, 2, 4 (1 and 3 are the ReLUs)."""

    def __init__(self, ch: int, mid_ch: int):
        super().__init__()
        self.add_module("0", HipConv2d(ch, mid_ch, 1))
        self.add_module("2", HipConv2d(mid_ch, mid_ch, 3, padding=1))
        self.add_module("4", HipConv2d(mid_ch, ch, 1))

    def __getitem__(self, i):
        return getattr(self, str(i))


class BaseBlock(nn.Module):
    def __init__(self, ch: int, mid_ch: int) -> None:
        super().__init__()
        self.conv = _ConvSeq(ch, mid_ch)

    def forward(self, x, affine=None):
        y = self.conv[0](x, act="relu")
        y = self.conv[2](y, act="relu")
        return self.conv[4](y, res=x, affine=affine)


class ResidualBottleneckBlocks(nn.Module):
    def __init__(self, ch: int, mid_ch: int, num_blocks: int = 3, res_in_res: bool = False):
        super().__init__()
        assert not res_in_res, "res_in_res is off in every CRDR config"
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, f"block{i}", BaseBlock(ch, mid_ch))

    def _chain(self, affine: bool) -> "CH.ChainSpec":
        """the whole stack as one hand-scheduled autograd node (crdr_amd/hip/chain.py)"""
        key = "_chain_aff" if affine else "_chain_plain"
        sp = self.__dict__.get(key)
        if sp is None:
            units = []
            for i in range(self.num_blocks):
                c = getattr(self, f"block{i}").conv
                units.append(CH.Unit([CH.Layer(c[0], "relu"), CH.Layer(c[2], "relu"), CH.Layer(c[4], None)], residual=True))
            sp = CH.ChainSpec([units], affine=affine, name="bottleneck")
            self.__dict__[key] = sp
        return sp

    def forward(self, x, affine=None):
        s, t = affine if affine is not None else (None, None)
        return CH.run_chain(x, self._chain(affine is not None), s, t)[0]


class BetaCondBaseBlock(nn.Module):
    """The three beta projections are 1x1 convs of the [1, cond_ch, 1, 1] conditioning vector; their outputs are
    per-channel vectors added after each ReLU / after the last 1x1 (epilogue VEC2)."""

    def __init__(self, ch: int, mid_ch: int, cond_ch: int) -> None:
        super().__init__()
        self.conv = _ConvSeq(ch, mid_ch)
        self.proj_1 = HipConv2d(cond_ch, mid_ch, 1)
        self.proj_2 = HipConv2d(cond_ch, mid_ch, 1)
        self.proj_3 = HipConv2d(cond_ch, ch, 1)

    def forward(self, x, cond_feat, affine=None):
        p1 = self.proj_1(cond_feat).reshape(-1)
        p2 = self.proj_2(cond_feat).reshape(-1)
        p3 = self.proj_3(cond_feat).reshape(-1)
        y = self.conv[0](x, act="relu", vec2=p1)
        y = self.conv[2](y, act="relu", vec2=p2)
        return self.conv[4](y, vec2=p3, res=x, affine=affine)


class BetaCondResidualBottleneckBlocks(nn.Module):
    def __init__(self, ch: int, mid_ch: int, cond_ch: int, num_blocks: int = 3, res_in_res: bool = False):
        super().__init__()
        assert not res_in_res
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, f"block{i}", BetaCondBaseBlock(ch, mid_ch, cond_ch))

    def _chain(self, affine: bool) -> "CH.ChainSpec":
        key = "_chain_aff" if affine else "_chain_plain"
        sp = self.__dict__.get(key)
        if sp is None:
            units = []
            for i in range(self.num_blocks):
                c = getattr(self, f"block{i}").conv
                units.append(CH.Unit([CH.Layer(c[0], "relu", 3 * i), CH.Layer(c[2], "relu", 3 * i + 1), CH.Layer(c[4], None, 3 * i + 2)],
                                     residual=True))
            sp = CH.ChainSpec([units], affine=affine, nvec=3 * self.num_blocks, name="beta_bottleneck")
            self.__dict__[key] = sp
        return sp

    def forward(self, x, cond_feat, affine=None):
        # all projections of the stack in one grouped launch per direction (they read the same conditioning vector)
        projs = [p for i in range(self.num_blocks) for p in (getattr(self, f"block{i}").proj_1, getattr(self, f"block{i}").proj_2,
                                                             getattr(self, f"block{i}").proj_3)]
        vecs = [v.reshape(-1) for v in HF.linear_group(cond_feat, projs)]
        s, t = affine if affine is not None else (None, None)
        return CH.run_chain(x, self._chain(affine is not None), s, t, vecs)[0]
