"""Channel-autoregressive ("Charm") context model of Minnen & Singh (ICIP 2020) as configured by CRDR
(src/models/subnet/context_model/minnen20_charm_context_model.py:26-240): 10 slices of 32 channels; for slice i
the mean / scale / LRP transforms (conv5-ReLU-conv5-ReLU-conv3) see the hyper-prior output plus the FIRST
min(i, 5) already decoded slices.

HIP mapping per slice: 9 fused conv launches (+ one fused Gaussian-conditional launch that yields the STE
latent, both likelihoods and both bit sums, + one LRP launch)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from crdr_amd.hip import functional as HF
from crdr_amd.models.layer.hip_layers import HipConv2d
from crdr_amd.utils.registry import CONTEXTMODEL_REGISTRY

from .base_context_model import BaseContextModel


class _Seq(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.add_module("0", HipConv2d(in_ch, 224, 5, padding=2))
        self.add_module("2", HipConv2d(224, 128, 5, padding=2))
        self.add_module("4", HipConv2d(128, out_ch, 3, padding=1))


class SliceTransform(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, actv: str = "relu"):
        super().__init__()
        assert actv == "relu"
        self.model = _Seq(in_ch, out_ch)

    def forward(self, x):
        x = getattr(self.model, "0")(x, act="relu")
        x = getattr(self.model, "2")(x, act="relu")
        return getattr(self.model, "4")(x)


@CONTEXTMODEL_REGISTRY.register()
class Minnen20CharmContextModel(BaseContextModel):
    def __init__(self, num_slices: int, bottleneck_y: int, hyper_out_ch: int, max_support_slices: int = 5,
                 slice_transform_kwargs: Dict = {}, crop_gaussian_params: bool = False) -> None:
        super().__init__()
        assert bottleneck_y % num_slices == 0
        assert max_support_slices == -1 or 1 <= max_support_slices <= num_slices
        assert not crop_gaussian_params
        slice_ch = bottleneck_y // num_slices
        hm = hyper_out_ch // 2
        self.slice_ch, self.num_slices, self.max_support_slices = slice_ch, num_slices, max_support_slices
        self.mean_slice_transforms = nn.ModuleList()
        self.scale_slice_transforms = nn.ModuleList()
        self.lrp_slice_transforms = nn.ModuleList()
        for i in range(num_slices):
            ns = i if max_support_slices == -1 else min(i, max_support_slices)
            sup = slice_ch * ns
            self.mean_slice_transforms.append(SliceTransform(sup + hm, slice_ch, **slice_transform_kwargs))
            self.scale_slice_transforms.append(SliceTransform(sup + hm, slice_ch, **slice_transform_kwargs))
            self.lrp_slice_transforms.append(SliceTransform(sup + hm + slice_ch, slice_ch, **slice_transform_kwargs))

    def _support(self, hats: List[Tensor]) -> List[Tensor]:
        return hats if self.max_support_slices < 0 else hats[: self.max_support_slices]

    @torch.no_grad()
    def reconstruct_latent(self, y: Tensor, h_mu: Tensor, entropy_model_y) -> Tensor:
        """y_hat only (the no-grad high-rate pass of stage 3 needs nothing else): y_hat_i = round(y_i - mu_i) + mu_i, then
        the LRP correction, do not depend on the scale transforms or on any likelihood, so those are not evaluated.
        Bit-identical to the y_hat of forward()."""
        ys = torch.chunk(y, self.num_slices, dim=1)
        hats: List[Tensor] = []
        ones = None
        for i, ysl in enumerate(ys):
            mean_support = torch.cat([h_mu] + self._support(hats), dim=1)
            mu = self.mean_slice_transforms[i](mean_support)
            if ones is None:
                ones = torch.ones_like(mu)
            yh = entropy_model_y.forward_split(ysl, mu, ones, is_train=False, want_bits=False, want_lik=False)[0]
            if getattr(self, "record_symbols", None) is not None:
                self.record_symbols.append(torch.round(yh.detach() - mu.detach()))
            z = self.lrp_slice_transforms[i](torch.cat([mean_support, yh], dim=1))
            hats.append(HF.lrp(yh, z))
        return torch.cat(hats, dim=1)

    def forward(self, y: Tensor, hyper_out: Tensor, entropy_model_y, is_train: bool, calc_q_likelihood: bool = True,
                noise: Tensor = None, want_lik: bool = True, bits_out: Dict = None):
        """-> (y_hat, y_likelihood, y_q_likelihood) like the reference; the per-image bit sums produced by the
        fused kernel are returned through `bits_out["y"], bits_out["y_q"]` when a dict is passed (nothing that carries
        an autograd graph is kept on the module)."""
        ys = torch.chunk(y, self.num_slices, dim=1)
        h_mu, h_sc = torch.chunk(hyper_out, 2, dim=1)
        ns = None if noise is None else torch.chunk(noise, self.num_slices, dim=1)
        hats, liks, qliks = [], [], []
        bits = bits_q = None
        for i, ysl in enumerate(ys):
            sup = self._support(hats)
            mean_support = torch.cat([h_mu] + sup, dim=1)
            scale_support = torch.cat([h_sc] + sup, dim=1)
            mu = self.mean_slice_transforms[i](mean_support)
            sigma = self.scale_slice_transforms[i](scale_support)
            yh, lik, b, bq, lq = entropy_model_y.forward_split(ysl, mu, sigma, is_train=is_train,
                                                              noise=None if ns is None else ns[i], want_bits=True,
                                                              want_lik=want_lik)
            if getattr(self, "record_symbols", None) is not None:  # parity tests read the rounding decisions
                self.record_symbols.append(torch.round(yh.detach() - mu.detach()))
            bits = b if bits is None else bits + b
            bits_q = bq if bits_q is None else bits_q + bq
            liks.append(lik)
            qliks.append(lq)
            z = self.lrp_slice_transforms[i](torch.cat([mean_support, yh], dim=1))
            hats.append(HF.lrp(yh, z))
        if bits_out is not None:
            bits_out["y"], bits_out["y_q"] = bits, bits_q
        y_hat = torch.cat(hats, dim=1)
        if not want_lik:
            return (y_hat, None, None) if calc_q_likelihood else (y_hat, None)
        y_lik = torch.cat(liks, dim=1)
        if calc_q_likelihood:
            return y_hat, y_lik, torch.cat(qliks, dim=1)
        return y_hat, y_lik

    # ---- codec paths (GPU transforms, host rANS)
    @torch.no_grad()
    def forward_compress(self, y: Tensor, hyper_out: Tensor, entropy_model_y) -> Tuple[List[bytes], Tensor, Tensor]:
        ys = torch.chunk(y, self.num_slices, dim=1)
        h_mu, h_sc = torch.chunk(hyper_out, 2, dim=1)
        hats, liks, mus, sigmas = [], [], [], []
        for i, ysl in enumerate(ys):
            sup = self._support(hats)
            mean_support = torch.cat([h_mu] + sup, dim=1)
            scale_support = torch.cat([h_sc] + sup, dim=1)
            mu = self.mean_slice_transforms[i](mean_support)
            sigma = self.scale_slice_transforms[i](scale_support)
            mus.append(mu)
            sigmas.append(sigma)
            yh, lik = entropy_model_y.forward_split(ysl, mu, sigma, is_train=False)
            z = self.lrp_slice_transforms[i](torch.cat([mean_support, yh], dim=1))
            hats.append(HF.lrp(yh, z))
            liks.append(lik)
        y_hat, y_lik = torch.cat(hats, 1), torch.cat(liks, 1)
        y_mean, y_scale = torch.cat(mus, 1), torch.cat(sigmas, 1)
        indexes = entropy_model_y.build_indexes(y_scale)
        y_str = entropy_model_y.compress(y, indexes=indexes, means=y_mean)
        return y_str, y_hat, y_lik

    @torch.no_grad()
    def forward_decompress(self, y_str: bytes, hyper_out: Tensor, entropy_model_y) -> Tuple[Tensor, Tensor]:
        from crdr_amd.codec import rans
        cdf = entropy_model_y._quantized_cdf.cpu().numpy()
        sizes = entropy_model_y._cdf_length.cpu().numpy()
        offs = entropy_model_y._offset.cpu().numpy()
        dec = rans.RansDecoder()
        dec.set_stream(y_str)
        h_mu, h_sc = torch.chunk(hyper_out, 2, dim=1)
        hats, syms = [], []
        for i in range(self.num_slices):
            sup = self._support(hats)
            mean_support = torch.cat([h_mu] + sup, dim=1)
            scale_support = torch.cat([h_sc] + sup, dim=1)
            mu = self.mean_slice_transforms[i](mean_support)
            sigma = self.scale_slice_transforms[i](scale_support)
            idx = entropy_model_y.build_indexes(sigma)
            vals = dec.decode_stream(idx.cpu().reshape(-1).int().numpy(), cdf, sizes, offs)
            sym = torch.from_numpy(vals).view(sigma.shape).to(sigma.device)
            yh = entropy_model_y.dequantize(sym, mu)
            z = self.lrp_slice_transforms[i](torch.cat([mean_support, yh], dim=1))
            hats.append(HF.lrp(yh, z))
            syms.append(sym)
        return torch.cat(hats, 1), torch.cat(syms, 1).int()
