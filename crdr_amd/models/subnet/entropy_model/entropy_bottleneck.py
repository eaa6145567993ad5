"""Factorised prior for the hyper-latent z.

The reference subclasses compressai 1.2.4's EntropyBottleneck (src/models/subnet/entropy_model/
entropy_bottleneck.py:13-30); that package is not available here, so the module is restated with the same
parameters (`_matrix{i}`, `_bias{i}`, `_factor{i}`, `quantiles`, buffers `target`, `_quantized_cdf`, `_offset`,
`_cdf_length`), initialisation and semantics. Likelihood / STE rounding / bit sums run in one HIP kernel.
PARITY UNPINNED against compressai itself (see oracle/crdr_oracle.py header)."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from crdr_amd.hip import functional as HF
from crdr_amd.utils.registry import ENTROPYMODEL_REGISTRY


@ENTROPYMODEL_REGISTRY.register()
class EntropyBottleneck(nn.Module):
    def __init__(self, channels: int, tail_mass: float = 1e-9, init_scale: float = 10, filters: Tuple[int, ...] = (3, 3, 3, 3),
                 likelihood_bound: float = 1e-9, entropy_coder_precision: int = 16):
        super().__init__()
        assert tuple(filters) == (3, 3, 3, 3), "the HIP kernel is specialised for filters (3,3,3,3)"
        self.channels, self.filters = int(channels), tuple(filters)
        self.init_scale, self.tail_mass = float(init_scale), float(tail_mass)
        self.likelihood_bound = float(likelihood_bound)
        self.entropy_coder_precision = int(entropy_coder_precision)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = math.log(math.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), init)))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.rand(channels, f[i + 1], 1) - 0.5))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        q = torch.tensor([-self.init_scale, 0.0, self.init_scale])
        self.quantiles = nn.Parameter(q.repeat(channels, 1, 1))
        t = math.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.tensor([-t, 0.0, t]))
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    # ---- parameter packing for the kernel: [C, 58] = L0{M3 b3 f3} L1..3{M9 b3 f3} L4{M3 b1}
    def packed_params(self) -> Tensor:
        c = self.channels
        parts = []
        for i in range(5):
            parts.append(getattr(self, f"_matrix{i}").reshape(c, -1))
            parts.append(getattr(self, f"_bias{i}").reshape(c, -1))
            if i < 4:
                parts.append(getattr(self, f"_factor{i}").reshape(c, -1))
        return torch.cat(parts, dim=1)

    def _get_medians(self) -> Tensor:
        return self.quantiles[:, :, 1:2]

    def forward(self, x: Tensor, is_train: bool = True, noise: Tensor = None, want_bits: bool = False):
        """-> (x_hat, likelihood).  Training: likelihood of x + U(-1/2, 1/2), output = x + noise (compressai's
        'noise' quantisation).  The STE subclass replaces the output by the rounded latent."""
        med = self._get_medians().detach().reshape(-1)
        if is_train and noise is None:
            noise = torch.rand(x.shape, device=x.device).contiguous(memory_format=torch.channels_last) - 0.5
        zhat, lik, bits = HF.entropy_bottleneck(x, self.packed_params(), med, noise if is_train else None, self.likelihood_bound)
        out = (x + noise) if is_train else zhat
        return (out, lik, bits) if want_bits else (out, lik)

    def loss(self) -> Tensor:
        """sum |logits_cdf(quantiles) - target| with the network parameters detached (aux optimiser objective)."""
        from crdr_amd.models.subnet.entropy_model.eb_aux import eb_aux_loss
        return eb_aux_loss(self)

    # ---- codec tables (host side, once per model)
    @torch.no_grad()
    def _logits_cumulative_host(self, x: Tensor) -> Tensor:
        h = x
        for i in range(5):
            h = torch.matmul(torch.nn.functional.softplus(getattr(self, f"_matrix{i}").detach().cpu()), h) + getattr(self, f"_bias{i}").detach().cpu()
            if i < 4:
                h = h + torch.tanh(getattr(self, f"_factor{i}").detach().cpu()) * torch.tanh(h)
        return h

    @torch.no_grad()
    def update(self, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        from crdr_amd.codec.tables import pmf_to_cdf_table
        qt = self.quantiles.detach().cpu()
        med = qt[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - qt[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(qt[:, 0, 2] - med).int(), min=0)
        start = med - minima
        length = maxima + minima + 1
        mx = int(length.max())
        samples = torch.arange(mx)[None, :] + start[:, None, None]
        lo = self._logits_cumulative_host(samples - 0.5)
        up = self._logits_cumulative_host(samples + 0.5)
        sign = -torch.sign(lo + up)
        pmf = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))[:, 0, :]
        tail = (torch.sigmoid(lo[:, 0, :1]) + torch.sigmoid(-up[:, 0, -1:]))[:, 0]
        table = pmf_to_cdf_table(pmf.numpy(), tail.numpy(), length.numpy(), mx, self.entropy_coder_precision)
        dev = self.quantiles.device
        self._quantized_cdf = torch.from_numpy(table).to(dev)
        self._cdf_length = (length + 2).int().to(dev)
        self._offset = (-minima).int().to(dev)
        return True

    @torch.no_grad()
    def quantize_symbols(self, x: Tensor) -> Tensor:
        med = self._get_medians().detach().reshape(1, -1, 1, 1).to(x.device)
        return torch.round(x - med).int()

    @torch.no_grad()
    def dequantize(self, symbols: Tensor) -> Tensor:
        med = self._get_medians().detach().reshape(1, -1, 1, 1).to(symbols.device)
        return symbols.float() + med

    @torch.no_grad()
    def compress(self, x: Tensor) -> List[bytes]:
        from crdr_amd.codec import rans
        sym = self.quantize_symbols(x).cpu()
        n, c, h, w = sym.shape
        idx = torch.arange(c, dtype=torch.int32).view(1, c, 1, 1).expand(n, c, h, w)
        cdf, sizes, offs = self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()
        return [rans.encode_with_indexes(sym[i].reshape(-1).numpy(), idx[i].reshape(-1).numpy(), cdf, sizes, offs) for i in range(n)]

    def compress_symbols(self, sym) -> List[bytes]:
        """host half of compress(): sym = int32 numpy [N, C, H, W] (quantize_symbols, copied to the host by the caller)"""
        from crdr_amd.codec import rans
        n, c, h, w = sym.shape
        idx = np.broadcast_to(np.arange(c, dtype=np.int32).reshape(c, 1, 1), (c, h, w)).reshape(-1)
        cdf, sizes, offs = self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()
        return [rans.encode_with_indexes(sym[i].reshape(-1), idx, cdf, sizes, offs) for i in range(n)]

    @torch.no_grad()
    def decompress(self, strings: List[bytes], size: Tuple[int, int]) -> Tensor:
        """-> integer symbols + medians (the caller's `dequantize` convention differs between reference models:
        hyperprior_charm_model.py:137-138 calls dequantize on the result, so symbols are returned here)."""
        from crdr_amd.codec import rans
        c = self.channels
        h, w = size
        idx = torch.arange(c, dtype=torch.int32).view(c, 1, 1).expand(c, h, w).reshape(-1).numpy()
        cdf, sizes, offs = self._quantized_cdf.cpu().numpy(), self._cdf_length.cpu().numpy(), self._offset.cpu().numpy()
        outs = [torch.from_numpy(rans.decode_with_indexes(s, idx, cdf, sizes, offs)).view(c, h, w) for s in strings]
        return torch.stack(outs, 0)


@ENTROPYMODEL_REGISTRY.register()
class SteEntropyBottleneck(EntropyBottleneck):
    """Noise for the rate estimate, straight-through rounding for what the decoder sees
    (entropy_bottleneck.py:18-30)."""

    def forward(self, x: Tensor, is_train: bool = True, noise: Tensor = None, want_bits: bool = False):
        med = self._get_medians().detach().reshape(-1)
        if is_train and noise is None:
            noise = torch.rand(x.shape, device=x.device).contiguous(memory_format=torch.channels_last) - 0.5
        zhat, lik, bits = HF.entropy_bottleneck(x, self.packed_params(), med, noise if is_train else None, self.likelihood_bound)
        return (zhat, lik, bits) if want_bits else (zhat, lik)
