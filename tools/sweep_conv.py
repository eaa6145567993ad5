"""Time every tile configuration x split depth of crdr_conv2d (and crdr_conv2d_wgrad) on a few representative stage-3
shapes and print the table -- the tool behind kernel experiments (compare two builds with CRDR_HIP_LIB=<other .so>).

Usage: python tools/sweep_conv.py [--bs 16] [--top 4] [--wgrad] [--shapes name,name]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402

# name: Cin, H, Cout, k, stride, transposed, flags
SHAPES = {
    "dec128k3": (128, 128, 128, 3, 1, 0, 3),
    "dec64k3": (128, 64, 128, 3, 1, 0, 3),
    "enc96k3": (96, 128, 96, 3, 1, 0, 3),
    "enc96k3@64": (96, 64, 96, 3, 1, 0, 3),
    "dec256k1": (256, 128, 128, 1, 1, 0, 3),
    "enc192k1@128": (192, 128, 96, 1, 1, 0, 3),
    "dec128to256k1@128": (128, 128, 256, 1, 1, 0, 17),
    "enc192k1@64": (192, 64, 96, 1, 1, 0, 3),
    "enc96to192k1@64": (96, 64, 192, 1, 1, 0, 17),
    "dec128to256k1@64": (128, 64, 256, 1, 1, 0, 17),
    "enc96to192k1@128": (96, 128, 192, 1, 1, 0, 17),
    "k3_128to64": (128, 128, 64, 3, 1, 0, 3),
    "k3_128to96": (128, 128, 96, 3, 1, 0, 3),
    "k3_128to192": (128, 128, 192, 3, 1, 0, 3),
    "k3_128to256": (128, 128, 256, 3, 1, 0, 3),
    "k3_128to320": (128, 64, 320, 3, 1, 0, 3),
    "D3to64": (3, 256, 64, 3, 1, 0, 5),
    "stem3": (3, 256, 192, 5, 2, 0, 1),
    "up3T": (256, 64, 256, 5, 2, 1, 1),
    "enc5s2": (192, 128, 192, 5, 2, 0, 1),
    "hoist4256": (320, 16, 4256, 5, 1, 0, 0),
    "hoist4256T": (4256, 16, 320, 5, 1, 1, 0),
    "charm480": (480, 16, 224, 5, 1, 0, 3),
    "charm224": (224, 16, 128, 5, 1, 0, 3),
    "charm128": (128, 16, 32, 3, 1, 0, 1),
    "nlam160": (160, 16, 160, 3, 1, 0, 3),
    "D256s2": (256, 64, 256, 3, 2, 0, 5),
    "gdn192": (192, 128, 192, 1, 1, 0, 0),     # the GDN channel mix and its gamma gradient (gdn.hip)
    "gdn192@32": (192, 32, 192, 1, 1, 0, 0),
}


def timeit(fn, iters=3):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--top", type=int, default=4)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--dump", action="store_true", help="print every unsplit tile configuration's TFLOP/s (cost-model fitting)")
    ap.add_argument("--bf16x3", action="store_true", help="time the split-bf16 kernels (precision: bf16x3)")
    ap.add_argument("--bf16x6", action="store_true", help="time the fp32-equivalent split-bf16 kernels (precision: bf16x6)")
    a = ap.parse_args()
    ops.MATRIX_BF16X3 = a.bf16x3
    ops.MATRIX_BF16X6 = a.bf16x6
    dev = torch.device("cuda:0")
    lib = L.load()
    ncfg = lib.crdr_conv2d_num_configs()
    nw = lib.crdr_conv2d_wgrad_num_configs()
    names = a.shapes.split(",") if a.shapes else list(SHAPES)
    print(f"lib {L.LIB_PATH}  conv configs {ncfg}  wgrad configs {nw}")
    for name in names:
        ci, h, co, k, s, tr, flags = SHAPES[name]
        p = k // 2
        oh = ops.conv_out_size(h, k, s, p, bool(tr), out_pad=(1 if (tr and s == 2) else 0))
        x = torch.randn(a.bs, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        x, _ = ops.nhwc(x)
        w = torch.randn(*((ci, co, k, k) if tr else (co, ci, k, k)), device=dev) * 0.02
        wf = ops.pack_weight(w, transpose=bool(tr))
        bias = torch.randn(co, device=dev)
        resid = torch.randn(a.bs, co, oh, oh, device=dev).contiguous(memory_format=torch.channels_last) if flags & 16 else None
        flops = 2.0 * a.bs * (h * h if tr else oh * oh) * ci * co * k * k
        res = []
        for c in range(ncfg):
            for ls in range(5):
                algo = (c + 1) | (ls << 8)
                try:
                    fn = lambda: ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh), bias=bias, res=resid, flags=flags, algo=algo)  # noqa: E731
                    t = timeit(fn)
                except L.CrdrHipError:
                    continue
                res.append((t, c, 1 << ls))
        if a.dump:
            print(f"{name:14s} unsplit TF by cfg: " + " ".join(f"{c}:{flops / t / 1e12:.0f}" for t, c, sp in sorted(res, key=lambda r: r[1]) if sp == 1), flush=True)
        res.sort()
        stream = []
        for v, algo in enumerate(ops._stream_ids()):
            try:
                t = timeit(lambda: ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh), bias=bias, res=resid, flags=flags, algo=algo))
            except L.CrdrHipError:
                continue
            stream.append((t, v))
        t0 = timeit(lambda: ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh), bias=bias, res=resid, flags=flags))
        print(f"{name:14s} heuristic {t0 * 1e6:8.1f} us {flops / t0 / 1e12:6.1f} TF | " +
              "  ".join(f"cfg{c}/s{sp} {t * 1e6:.1f}us {flops / t / 1e12:.1f}TF" for t, c, sp in res[:a.top]) +
              (" | stream " + "  ".join(f"v{v} {t * 1e6:.1f}us {flops / t / 1e12:.1f}TF" for t, v in stream) if stream else ""), flush=True)
        if a.wgrad and not tr:
            dy = torch.randn(a.bs, co, oh, oh, device=dev).contiguous(memory_format=torch.channels_last)
            g = torch.empty(co, ci, k, k, device=dev)
            resw = []
            for c in range(nw):
                for ls in range(9):
                    algo = (c + 1) | (ls << 8)
                    try:
                        t = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, g, (k, k), s, p, False, algo=algo))
                    except (L.CrdrHipError, AssertionError):
                        continue
                    resw.append((t, c, 1 << ls))
            resw.sort()
            print(f"{'':14s} wgrad | " + "  ".join(f"cfg{c}/s{sp} {t * 1e6:.1f}us {flops / t / 1e12:.1f}TF" for t, c, sp in resw[:a.top]), flush=True)


if __name__ == "__main__":
    main()
