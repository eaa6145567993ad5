"""Generalised divisive normalisation (GDN / inverse GDN) with the parameters, buffers and state-dict keys of
`compressai.layers.GDN` 1.2.4 (used by the reference's Balle18 / Cheng20 ablation transforms:
src/models/subnet/autoencoder/balle18_autoencoder.py:16-20,37-41; src/models/layer/cheng_resblock.py:8-15):

    y_i = x_i / sqrt(beta_i + sum_j gamma_ij x_j^2)        (inverse=True: multiply instead)

`beta` / `gamma` hold the *stored* values of the NonNegativeParametrizer (sqrt(max(v + pedestal, pedestal))); the
lower bounds, the squaring and the LowerBound gradient rule are applied inside crdr_gdn_{fwd,bwd}.  Not on the CRDR hot
path (the ELIC transforms use ReLU bottlenecks): an optional registered op."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from crdr_amd.hip import functional as HF
from crdr_amd.hip import lib as L
from crdr_amd.hip import ops


class _GdnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, beta, gamma, inverse: bool, beta_min: float, reparam_offset: float):
        lib = L.load()
        x, ldx = ops.nhwc(x)
        n, c, h, w = x.shape
        y = ops.empty_nhwc(n, c, h, w, x.device)
        d = L.GdnDesc(M=n * h * w, C=c, ldx=ldx, ldy=ops.ld_for(c), inverse=int(inverse), beta_min=beta_min, reparam_offset=reparam_offset)
        nb = lib.crdr_gdn_workspace(C.byref(d), 0)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
        p = (ws.data_ptr() + 255) // 256 * 256
        L.check(lib.crdr_gdn_fwd(C.byref(d), x.data_ptr(), beta.data_ptr(), gamma.data_ptr(), y.data_ptr(), p, nb, ops._stream()), "gdn_fwd")
        ctx.save_for_backward(x, beta, gamma)
        ctx.cfg = (inverse, beta_min, reparam_offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, beta, gamma = ctx.saved_tensors
        inverse, beta_min, reparam_offset = ctx.cfg
        lib = L.load()
        x, ldx = ops.nhwc(x)
        dy, lddy = ops.nhwc(dy)
        n, c, h, w = x.shape
        dx = ops.empty_nhwc(n, c, h, w, x.device)
        d = L.GdnDesc(M=n * h * w, C=c, ldx=ldx, ldy=ops.ld_for(c), inverse=int(inverse), beta_min=beta_min, reparam_offset=reparam_offset)
        nb = lib.crdr_gdn_workspace(C.byref(d), 1)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
        p = (ws.data_ptr() + 255) // 256 * 256
        gb, gg = HF._grad_slot(beta), HF._grad_slot(gamma)
        L.check(lib.crdr_gdn_bwd(C.byref(d), x.data_ptr(), beta.data_ptr(), gamma.data_ptr(), dy.data_ptr(), lddy, dx.data_ptr(),
                                 ops.ld_for(c), gb.data_ptr(), gg.data_ptr(), p, nb, ops._stream()), "gdn_bwd")
        return dx, None, None, None, None, None


class GDN(nn.Module):
    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6, gamma_init: float = 0.1):
        super().__init__()
        assert in_channels % 4 == 0, "GDN on the HIP path needs a channel count that is a multiple of 4"
        self.inverse, self.beta_min = bool(inverse), float(beta_min)
        self.reparam_offset = 2.0 ** -18
        ped = self.reparam_offset ** 2
        self.beta = nn.Parameter(torch.sqrt(torch.clamp(torch.ones(in_channels) + ped, min=ped)))
        self.gamma = nn.Parameter(torch.sqrt(torch.clamp(gamma_init * torch.eye(in_channels) + ped, min=ped)))
        # compressai keeps these as buffers of the two parametrizers: same keys so that its checkpoints load strictly
        for name, bound in (("beta_reparam", (self.beta_min + ped) ** 0.5), ("gamma_reparam", self.reparam_offset)):
            m = nn.Module()
            m.register_buffer("pedestal", torch.tensor([ped]))
            lb = nn.Module()
            lb.register_buffer("bound", torch.tensor([float(bound)]))
            m.add_module("lower_bound", lb)
            self.add_module(name, m)

    def forward(self, x):
        return _GdnFn.apply(x, self.beta, self.gamma, self.inverse, self.beta_min, self.reparam_offset)
