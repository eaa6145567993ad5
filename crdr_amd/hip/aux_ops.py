"""Small autograd wrappers that do not fit functional.py's conv/entropy groups."""
from __future__ import annotations

import torch

from . import lib as L
from . import ops


class _EbQuantileLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quantiles, params, target):
        lib = L.load()
        ops._require_gpu(quantiles)
        q = quantiles.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=q.device)
        dq = torch.empty_like(q)
        L.check(lib.crdr_eb_quantile_loss(q.data_ptr(), params.contiguous().data_ptr(), target.contiguous().data_ptr(),
                                          q.shape[0], loss.data_ptr(), dq.data_ptr(), ops._stream()), "eb_quantile_loss")
        ctx.save_for_backward(dq)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return dq * g, None, None


def eb_quantile_loss(quantiles, params, target):
    return _EbQuantileLoss.apply(quantiles, params, target)
