"""Per-shape timing of the conv kernels at the stage-3 training shapes (bs 16, 256x256 crops).
Usage: python tools/bench_kernels.py [--bs 16] [--out gpurun_out/kernels.txt]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.hip import ops  # noqa: E402

# name, Cin, H(in), Cout, k, stride, transposed, count-per-G-forward
SHAPES = [
    ("enc.stem 3->192 k5s2 @256", 3, 256, 192, 5, 2, 0),
    ("enc.b1 192->96 k1 @128", 192, 128, 96, 1, 1, 0),
    ("enc.b1 96->96 k3 @128", 96, 128, 96, 3, 1, 0),
    ("enc.b1 96->192 k1 @128", 96, 128, 192, 1, 1, 0),
    ("enc.conv2 192->192 k5s2 @128", 192, 128, 192, 5, 2, 0),
    ("enc.b2 96->96 k3 @64", 96, 64, 96, 3, 1, 0),
    ("enc.b2 192->96 k1 @64", 192, 64, 96, 1, 1, 0),
    ("enc.conv3 192->192 k5s2 @64", 192, 64, 192, 5, 2, 0),
    ("enc.conv4 192->320 k5s2 @32", 192, 32, 320, 5, 2, 0),
    ("nlam 160->160 k3 @16", 160, 16, 160, 3, 1, 0),
    ("nlam 320->160 k1 @16", 320, 16, 160, 1, 1, 0),
    ("henc 320->320 k3 @16", 320, 16, 320, 3, 1, 0),
    ("henc 320->256 k5s2 @16", 320, 16, 256, 5, 2, 0),
    ("hdec T192->192 k5s2 @4", 192, 4, 192, 5, 2, 1),
    ("hdec T192->256 k5s2 @8", 192, 8, 256, 5, 2, 1),
    ("hdec T256->320 k3 @16", 256, 16, 320, 3, 1, 1),
    ("charm 320->224 k5 @16", 320, 16, 224, 5, 1, 0),
    ("charm 480->224 k5 @16", 480, 16, 224, 5, 1, 0),
    ("charm 224->128 k5 @16", 224, 16, 128, 5, 1, 0),
    ("charm 128->32 k3 @16", 128, 16, 32, 3, 1, 0),
    ("dec.up1 T320->256 k5s2 @16", 320, 16, 256, 5, 2, 1),
    ("dec.up2 T256->256 k5s2 @32", 256, 32, 256, 5, 2, 1),
    ("dec.up3 T256->256 k5s2 @64", 256, 64, 256, 5, 2, 1),
    ("dec.up4 T256->3 k5s2 @128", 256, 128, 3, 5, 2, 1),
    ("dec.b 256->128 k1 @128", 256, 128, 128, 1, 1, 0),
    ("dec.b 128->128 k3 @128", 128, 128, 128, 3, 1, 0),
    ("dec.b 128->256 k1 @128", 128, 128, 256, 1, 1, 0),
    ("dec.b 128->128 k3 @64", 128, 64, 128, 3, 1, 0),
    ("D 3->64 k3 @256", 3, 256, 64, 3, 1, 0),
    ("D 64->64 k3s2 @256", 64, 256, 64, 3, 2, 0),
    ("D 64->128 k3 @128", 64, 128, 128, 3, 1, 0),
    ("D 128->128 k3s2 @128", 128, 128, 128, 3, 2, 0),
    ("D 256->256 k3s2 @64", 256, 64, 256, 3, 2, 0),
    ("D 512->512 k3s2 @32", 512, 32, 512, 3, 2, 0),
    ("D 512->1 k3 @16", 512, 16, 1, 3, 1, 0),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lines = []
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, ci, h, co, k, s, tr in SHAPES:
        p = k // 2
        oh = ops.conv_out_size(h, k, s, p, bool(tr), out_pad=(1 if (tr and s == 2) else 0))
        x = torch.randn(a.bs, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last) if ci % 4 == 0 else torch.randn(a.bs, ci, h, h, device=dev)
        x, _ = ops.nhwc(x)
        wshape = (ci, co, k, k) if tr else (co, ci, k, k)
        w = torch.randn(*wshape, device=dev) * 0.02
        wf = ops.pack_weight(w, transpose=bool(tr))
        wb = ops.pack_weight(w, transpose=not tr)
        y = ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh))
        dy = torch.randn_like(y)
        dy, _ = ops.nhwc(dy)
        g = torch.zeros_like(w)
        flops = 2.0 * a.bs * (h * h if tr else oh * oh) * ci * co * k * k
        t_f = timeit(lambda: ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh), out=y))
        dx = ops.conv2d_raw(dy, wb, ci, (k, k), s, p, not tr, (h, h))
        t_d = timeit(lambda: ops.conv2d_raw(dy, wb, ci, (k, k), s, p, not tr, (h, h), out=dx))
        if tr:
            t_w = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, g, (k, k), s, p, False))
        else:
            t_w = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, g, (k, k), s, p, False))
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        lines.append(f"{name:34s} GF={flops/1e9:8.2f}  fwd {t_f*1e6:8.1f}us {flops/t_f/1e12:6.1f}TF | dgrad {t_d*1e6:8.1f}us {flops/t_d/1e12:6.1f}TF | wgrad {t_w*1e6:8.1f}us {flops/t_w/1e12:6.1f}TF")
        print(lines[-1], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
