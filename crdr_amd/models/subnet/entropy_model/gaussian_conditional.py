"""Gaussian conditional entropy model with mean and scale (the reference subclasses compressai 1.2.4's
GaussianConditional: src/models/subnet/entropy_model/gaussian_conditional.py:10-24).  Restated: scale lower
bound 0.11, likelihood lower bound 1e-9, likelihood = Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), 64-level log-spaced
scale table for coding.  PARITY UNPINNED against compressai itself."""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from crdr_amd.hip import functional as HF
from crdr_amd.utils.registry import ENTROPYMODEL_REGISTRY

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(lo: float = SCALES_MIN, hi: float = SCALES_MAX, levels: int = SCALES_LEVELS) -> Tensor:
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


@ENTROPYMODEL_REGISTRY.register()
class GaussianMeanScaleConditional(nn.Module):
    def __init__(self, scale_bound: Optional[float] = None, tail_mass: float = 1e-9, likelihood_bound: float = 1e-9,
                 entropy_coder_precision: int = 16):
        super().__init__()
        self.scale_bound = 0.11 if scale_bound is None else float(scale_bound)
        self.tail_mass, self.likelihood_bound = float(tail_mass), float(likelihood_bound)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.register_buffer("scale_table", torch.Tensor())
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    def forward(self, y: Tensor, params: Tensor, is_train: bool = True, noise: Tensor = None, want_bits: bool = False):
        """-> (y_out, likelihood): training returns the noisy latent and its likelihood; eval the dequantised one."""
        mean, std = params.chunk(2, 1)
        if is_train and noise is None:
            noise = torch.rand(y.shape, device=y.device).contiguous(memory_format=torch.channels_last) - 0.5
        yhat, bn, bq, ln, lq = HF.gauss_cond(y, mean, std, noise if is_train else None, self.scale_bound, self.likelihood_bound, True)
        out, lik, bits = ((y + noise), ln, bn) if is_train else (yhat, lq, bq)
        return (out, lik, bits) if want_bits else (out, lik)

    # ---- codec side (host)
    @torch.no_grad()
    def update_scale_table(self, scale_table: Tensor, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        from scipy.stats import norm
        from crdr_amd.codec.tables import pmf_to_cdf_table, std_cdf
        st = scale_table.detach().float().cpu()
        mult = -norm.ppf(self.tail_mass / 2)
        center = torch.ceil(st * mult).int()
        length = 2 * center + 1
        mx = int(length.max())
        samples = torch.abs(torch.arange(mx).int() - center[:, None]).float()
        sc = st.unsqueeze(1)
        upper, lower = std_cdf((0.5 - samples) / sc), std_cdf((-0.5 - samples) / sc)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
        table = pmf_to_cdf_table(pmf.numpy(), tail[:, 0].numpy(), length.numpy(), mx, self.entropy_coder_precision)
        dev = self._offset.device
        self.scale_table = st.to(dev)
        self._quantized_cdf = torch.from_numpy(table).to(dev)
        self._cdf_length = (length + 2).int().to(dev)
        self._offset = (-center).int().to(dev)
        return True

    def host_tables(self):
        """(quantised CDFs, lengths, offsets) as numpy arrays, cached until the tables are rebuilt"""
        key = (self._quantized_cdf.data_ptr(), tuple(self._quantized_cdf.shape))
        c = self.__dict__.get("_host_tables")
        if c is None or c[0] != key:
            c = (key, (self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()))
            self.__dict__["_host_tables"] = c
        return c[1]

    @torch.no_grad()
    def build_indexes(self, scales: Tensor) -> Tensor:
        s = torch.clamp(scales, min=self.scale_bound)
        idx = torch.full(s.shape, len(self.scale_table) - 1, dtype=torch.int32, device=s.device)
        for v in self.scale_table[:-1].tolist():
            idx -= (s <= v).int()
        return idx

    @torch.no_grad()
    def symbols_and_indexes(self, y: Optional[Tensor], means: Optional[Tensor], scales: Tensor, out_sym: Optional[Tensor] = None,
                            out_idx: Optional[Tensor] = None) -> Tuple[Optional[Tensor], Tensor]:
        """One launch: int32 symbols round(y - means) and CDF indexes of `scales`, both [N, C, H, W] CONTIGUOUS (the order the
        host coder walks), from NHWC device tensors (crdr_gauss_symbols).  y = None: indexes only (decoder side)."""
        import ctypes as C
        from crdr_amd.hip import lib as L
        from crdr_amd.hip import ops
        sg, lds = ops.nhwc(scales)
        n, c, h, w = sg.shape
        dev = sg.device
        idx = out_idx if out_idx is not None else torch.empty((n, c, h, w), dtype=torch.int32, device=dev)
        sym = None
        yp = mp = None
        ldy = ldm = 0
        if y is not None:
            yy, ldy = ops.nhwc(y)
            mm, ldm = ops.nhwc(means)
            yp, mp = yy.data_ptr(), mm.data_ptr()
            sym = out_sym if out_sym is not None else torch.empty((n, c, h, w), dtype=torch.int32, device=dev)
        tab = self.scale_table.to(dev) if self.scale_table.device != dev else self.scale_table
        L.check(L.load().crdr_gauss_symbols(yp, ldy, mp, ldm, sg.data_ptr(), lds, tab.data_ptr(), tab.numel(), self.scale_bound, n, h * w, c,
                                            None if sym is None else sym.data_ptr(), idx.data_ptr(), ops._stream()), "gauss_symbols")
        return sym, idx

    @staticmethod
    def quantize(inputs: Tensor, mode: str, means: Optional[Tensor] = None) -> Tensor:
        assert mode in ("dequantize", "symbols")
        out = inputs.clone()
        if means is not None:
            out -= means
        out = torch.round(out)
        if mode == "dequantize":
            return out + means if means is not None else out
        return out.int()

    @staticmethod
    def dequantize(symbols: Tensor, means: Optional[Tensor] = None) -> Tensor:
        out = symbols.float()
        return out + means if means is not None else out

    @torch.no_grad()
    def compress(self, inputs: Tensor, indexes: Tensor, means: Optional[Tensor] = None) -> List[bytes]:
        from crdr_amd.codec import rans
        sym = self.quantize(inputs, "symbols", means).cpu()
        idx = indexes.cpu()
        cdf, sizes, offs = self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()
        return [rans.encode_with_indexes(sym[i].reshape(-1).numpy(), idx[i].reshape(-1).int().numpy(), cdf, sizes, offs)
                for i in range(sym.shape[0])]

    @torch.no_grad()
    def decompress(self, strings: List[bytes], indexes: Tensor, means: Optional[Tensor] = None) -> Tensor:
        from crdr_amd.codec import rans
        idx = indexes.cpu()
        cdf, sizes, offs = self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()
        outs = []
        for i, s in enumerate(strings):
            v = rans.decode_with_indexes(s, idx[i].reshape(-1).int().numpy(), cdf, sizes, offs)
            outs.append(torch.from_numpy(v).view(idx.shape[1:]))
        sym = torch.stack(outs, 0).to(indexes.device)
        return self.dequantize(sym, means)
