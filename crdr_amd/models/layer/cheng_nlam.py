"""Cheng et al. (CVPR 2020) simplified attention: x + trunk(x) * sigmoid(conv1x1(attn(x))) with conv-only
branches (src/models/layer/cheng_nlam.py:5-46).  The gate -- and the InterpChAtt that follows the module in the
CRDR transforms -- is the epilogue of the final 1x1 conv."""
from __future__ import annotations

import torch.nn as nn

from crdr_amd.hip import chain as CH

from .hip_layers import HipConv2d


class NLAMResBlock(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, padding_mode: str = "zeros"):
        super().__init__()
        assert padding_mode == "zeros" and in_ch == out_ch
        mid = out_ch // 2
        self.c1 = HipConv2d(in_ch, mid, 1)
        self.c2 = HipConv2d(mid, mid, 3, padding=1)
        self.c3 = HipConv2d(mid, out_ch, 1)

    def forward(self, x):
        y = self.c1(x, act="relu")
        y = self.c2(y, act="relu")
        return self.c3(y, res=x)


class _Seq3(nn.Module):
    def __init__(self, ch):
        super().__init__()
        for i in range(3):
            self.add_module(str(i), NLAMResBlock(ch, ch))

    def forward(self, x):
        for i in range(3):
            x = getattr(self, str(i))(x)
        return x


class ChengNLAM(nn.Module):
    def __init__(self, ch: int, padding_mode: str = "zeros"):
        super().__init__()
        self.trunk_block = _Seq3(ch)
        self.attention_block = _Seq3(ch)
        self.conv = HipConv2d(ch, ch, 1)

    def _chain(self) -> "CH.ChainSpec":
        """trunk and attention branches (three residual 1-3-1 bottlenecks each, both reading x) as ONE autograd node of
        grouped launches (crdr_amd/hip/chain.py)"""
        sp = self.__dict__.get("_chain_spec")
        if sp is None:
            def units(seq):
                return [CH.Unit([CH.Layer(b.c1, "relu"), CH.Layer(b.c2, "relu"), CH.Layer(b.c3, None)], residual=True)
                        for b in (getattr(seq, str(i)) for i in range(3))]
            sp = CH.ChainSpec([units(self.trunk_block), units(self.attention_block)], name="nlam")
            self.__dict__["_chain_spec"] = sp
        return sp

    def forward(self, x, affine=None):
        trunk, attn = CH.run_chain(x, self._chain())
        return self.conv(attn, gate=(x, trunk), affine=affine)
