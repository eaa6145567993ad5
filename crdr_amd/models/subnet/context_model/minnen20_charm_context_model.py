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

    # ---- noise: Philox state (seed, offset) on the device; the fused kernels draw U(-1/2, 1/2) from it and the backward
    # regenerates the same samples (crdr_gauss_cond_fwd2).  Seeded per rank by the trainer (seed_noise).
    def _philox(self, device) -> Tensor:
        st = getattr(self, "_philox_state", None)
        if st is None or st.device != device:
            seed = getattr(self, "_noise_seed", None)
            if seed is None:   # never seeded by a trainer: still one stream per rank
                from crdr_amd.trainer import dist as _D
                seed = (torch.initial_seed() + 7919 * (_D.rank() + 1)) & 0x7FFFFFFFFFFFFFFF
            st = torch.tensor([int(seed), int(getattr(self, "_noise_offset", 0))], dtype=torch.int64, device=device)
            self._philox_state = st
        return st

    def seed_noise(self, seed: int, offset: int = 0) -> None:
        self._noise_seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        self._noise_offset = int(offset)
        self._philox_state = None

    def noise_state(self) -> Dict:
        """(seed, offset) of the in-kernel noise generator, for the trainer's checkpoint: a resumed run continues the
        sequence instead of replaying it from offset 0.  Synchronises."""
        st = getattr(self, "_philox_state", None)
        if st is None:
            return {"seed": getattr(self, "_noise_seed", None), "offset": int(getattr(self, "_noise_offset", 0))}
        seed, off = st.tolist()
        return {"seed": int(seed), "offset": int(off)}

    def load_noise_state(self, state: Dict) -> None:
        if state and state.get("seed") is not None:
            self.seed_noise(state["seed"], state.get("offset", 0))

    def _record(self, run) -> None:
        """parity tests read the rounding decisions (round(y_hat_pre - mu) per slice)"""
        if getattr(self, "record_symbols", None) is not None:
            yp, mu = run.as_nchw(run.Ypre), run.as_nchw(run.MSL, 0, run.Cy)
            for i in range(self.num_slices):
                a, b = i * self.slice_ch, (i + 1) * self.slice_ch
                self.record_symbols.append(torch.round(yp[:, a:b] - mu[:, a:b]))

    @torch.no_grad()
    def reconstruct_latent(self, y: Tensor, h_mu: Tensor, entropy_model_y) -> Tensor:
        """y_hat only (the no-grad high-rate pass of stage 3 needs nothing else): y_hat_i = round(y_i - mu_i) + mu_i, then
        the LRP correction, do not depend on the scale transforms or on any likelihood, so those are not evaluated.
        Bit-identical to the y_hat of forward()."""
        from crdr_amd.hip import charm
        run = charm.charm_reconstruct(self, y, h_mu, entropy_model_y.scale_bound, entropy_model_y.likelihood_bound)
        self._record(run)
        return run.as_nchw(run.Yh)

    def forward(self, y: Tensor, hyper_out: Tensor, entropy_model_y, is_train: bool, calc_q_likelihood: bool = True,
                noise: Tensor = None, want_lik: bool = True, bits_out: Dict = None):
        """-> (y_hat, y_likelihood, y_q_likelihood) like the reference (:88-141); the per-image bit sums produced by the
        fused kernels are returned through `bits_out["y"], bits_out["y_q"]` when a dict is passed -- and ONLY those carry the
        rate gradient: the likelihood tensors are outputs of the fused node marked non-differentiable (a rate loss must be
        built from the bit sums, as models/comp_model does; -log2 of the returned likelihoods would train nothing).  The whole slice loop runs
        in crdr_amd.hip.charm (hoisted hyper-prior convs, grouped launches, hand-written backward)."""
        from crdr_amd.hip import charm
        yh, bn, bq, lik_n, lik_q, mu, sigma = charm.charm_forward(
            self, y, hyper_out, noise, self._philox(y.device) if (is_train and noise is None) else None,
            entropy_model_y.scale_bound, entropy_model_y.likelihood_bound, want_lik, is_train)
        if getattr(self, "record_symbols", None) is not None:
            for i in range(self.num_slices):
                a, b = i * self.slice_ch, (i + 1) * self.slice_ch
                # y_hat before the LRP correction = round(y - mu) + mu
                self.record_symbols.append(torch.round(y.detach()[:, a:b] - mu[:, a:b]))
        if bits_out is not None:
            bits_out["y"], bits_out["y_q"] = (bn if is_train else bq), bq
        if not want_lik:
            return (yh, None, None) if calc_q_likelihood else (yh, None)
        lik = lik_n if is_train else lik_q
        if calc_q_likelihood:
            return yh, lik, lik_q
        return yh, lik

    # ---- codec paths (GPU transforms, host rANS)
    codec_profile = None  # shared with the model's compress / decompress (wall-time split {charm, rans})

    def _tick(self, key, t0=None):
        import time
        if self.codec_profile is None:
            return 0.0
        torch.cuda.synchronize()
        t = time.perf_counter()
        if key is not None:
            self.codec_profile[key] = self.codec_profile.get(key, 0.0) + (t - t0)
        return t

    @torch.no_grad()
    def forward_compress_device(self, y: Tensor, hyper_out: Tensor, entropy_model_y):
        """GPU half of forward_compress (:143-187): all transforms, then ONE launch turns (y, mu, sigma) into the int32 symbols
        and CDF indexes in coder order.  -> (symbols, indexes, y_hat, likelihood), all on the device; nothing synchronises."""
        yh, _, _, _, lik_q, mu, sigma = self.forward_raw_eval(y, hyper_out, entropy_model_y)
        sym, idx = entropy_model_y.symbols_and_indexes(y, mu, sigma)
        return sym, idx, yh, lik_q

    @torch.no_grad()
    def forward_compress(self, y: Tensor, hyper_out: Tensor, entropy_model_y) -> Tuple[List[bytes], Tensor, Tensor]:
        from crdr_amd.codec import rans
        t = self._tick(None)
        sym, idx, yh, lik_q = self.forward_compress_device(y, hyper_out, entropy_model_y)
        t = self._tick("charm", t)
        cdf, sizes, offs = entropy_model_y.host_tables()
        sym, idx = sym.cpu().numpy(), idx.cpu().numpy()
        y_str = [rans.encode_with_indexes(sym[i].reshape(-1), idx[i].reshape(-1), cdf, sizes, offs) for i in range(sym.shape[0])]
        self._tick("rans", t)
        return y_str, yh, lik_q

    @torch.no_grad()
    def forward_raw_eval(self, y, hyper_out, entropy_model_y):
        from crdr_amd.hip import charm
        return charm.charm_forward(self, y, hyper_out, None, None, entropy_model_y.scale_bound, entropy_model_y.likelihood_bound,
                                   True, False)

    def _pinned_pair(self, shape):
        """Two pinned int32 host buffers of (at least) `shape` elements, kept per thread (decompress_many decodes several images
        concurrently, one thread and one stream each)."""
        import threading
        cache = self.__dict__.setdefault("_pin_cache", {})
        key = threading.get_ident()
        need = 1
        for d in shape:
            need *= int(d)
        pair = cache.get(key)
        if pair is None or pair[0].numel() < need:
            pair = cache[key] = (torch.empty(need, dtype=torch.int32, pin_memory=True), torch.empty(need, dtype=torch.int32, pin_memory=True))
        return pair

    @torch.no_grad()
    def forward_decompress(self, y_str: bytes, hyper_out: Tensor, entropy_model_y) -> Tuple[Tensor, Tensor]:
        """Decoder side (:189-240): the serial rANS decoder on the host alternates with the GPU transforms -- ms + 1 round
        trips (one per sequential slice, one for the whole tail) instead of one per slice."""
        from crdr_amd.codec import rans
        from crdr_amd.hip import charm
        cdf, sizes, offs = entropy_model_y.host_tables()
        dec = rans.RansDecoder()
        dec.set_stream(y_str)
        h_mu, h_sc = torch.chunk(hyper_out, 2, dim=1)
        run = charm.CharmRun(charm.plan_for(self), h_mu, h_sc)
        run.scale_bound, run.lik_bound = entropy_model_y.scale_bound, entropy_model_y.likelihood_bound
        run.hoist()
        sc = self.slice_ch
        mu_all, sg_all = run.as_nchw(run.MSL, 0, run.Cy), run.as_nchw(run.MSL, run.Cy, run.Cy)
        yh_all, yp_all = run.as_nchw(run.Yh), run.as_nchw(run.Ypre)
        dev = hyper_out.device
        syms = torch.empty((run.n, run.Cy, run.h, run.w), dtype=torch.int32, device=dev)
        # host <-> device staging of one stage (CDF indexes down, decoded symbols up) through pinned buffers, asynchronous
        # copies on the caller's stream: the host waits only for the event behind the index copy, and the symbol upload is in
        # flight while the next stage's launches are being issued
        biggest = max(len(st) for st in run.stages()) * sc
        pin_idx, pin_sym = self._pinned_pair((run.n, biggest, run.h, run.w))
        ev = torch.cuda.Event()
        t = self._tick(None)
        for st in run.stages():
            run.mean_scale(st)
            t = self._tick("charm", t)
            # the stream holds the symbols in (channel, row, column) order: the channels of a stage are consecutive in it
            a, b = st[0] * sc, (st[-1] + 1) * sc
            cnt = run.n * (b - a) * run.h * run.w
            _, idx = entropy_model_y.symbols_and_indexes(None, None, sg_all[:, a:b])
            hi, hs = pin_idx.view(-1)[:cnt], pin_sym.view(-1)[:cnt]
            hi.copy_(idx.view(-1), non_blocking=True)
            ev.record()
            ev.synchronize()
            dec.decode_stream_into(hi.numpy(), cdf, sizes, offs, hs.numpy())
            sym = torch.empty((run.n, b - a, run.h, run.w), dtype=torch.int32, device=dev)
            sym.view(-1).copy_(hs, non_blocking=True)
            syms[:, a:b] = sym
            v = entropy_model_y.dequantize(sym, mu_all[:, a:b])
            yh_all[:, a:b] = v
            yp_all[:, a:b] = v
            ev.record()   # the pinned symbol buffer is reused by the next stage: its upload must have been consumed by then
            t = self._tick("rans", t)
            run.lrp(st)
            ev.synchronize()
        self._tick("charm", t)
        return yh_all, syms
