"""Achieved HBM bandwidth of the GDN / IGDN op (crdr_gdn_fwd / crdr_gdn_bwd) at the Balle18 analysis sizes (bs 16, 192 channels):
algorithmic bytes = forward: read x, write y (8 B / element; x^2 and the norm are internal scratch); backward: read x, dy, write dx
(12 B / element).  Usage: python tools/gdn_bw.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.models.layer.gdn import GDN  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    out = {}
    for inverse in (False, True):
        m = GDN(192, inverse=inverse).to(dev)
        for hw in (128, 64, 32):
            x = torch.randn(16, 192, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            tf = timed(lambda: m(x.detach()))
            y = m(x)
            g = torch.randn_like(y)
            tb = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
            n = x.numel()
            flop = 2.0 * 16 * hw * hw * 192 * 192
            out[f"{'igdn' if inverse else 'gdn'} 16x192x{hw}x{hw}"] = {
                "fwd_us": round(tf * 1e6, 1), "fwd_GBps_algorithmic": round(8.0 * n / tf / 1e9, 1), "fwd_mix_TFLOPs": round(flop / tf / 1e12, 1),
                "bwd_us": round(tb * 1e6, 1), "bwd_GBps_algorithmic": round(12.0 * n / tb / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
