"""Winograd F(2x2, 3x3) kernel against the tuned implicit-GEMM kernel on the 3x3 stride-1 shapes of the stage-3 step (bs 16):
python tools/bench_wino.py [--cold]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402

SHAPES = [(128, 128, 128), (96, 96, 128), (64, 128, 128), (128, 64, 128), (128, 128, 64), (96, 96, 64), (128, 256, 64), (256, 128, 64),
          (256, 512, 32), (512, 256, 32), (128, 128, 32), (96, 96, 32), (160, 160, 16), (320, 320, 16)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--wino-only", action="store_true")
    ap.add_argument("--shapes", default=None, help="ci,co,hw;ci,co,hw;... instead of the built-in list")
    ap.add_argument("--k5", action="store_true", help="the 5x5 stride-1 shapes of the Charm at 16x16 instead")
    ap.add_argument("--k5s2", action="store_true", help="the 5x5 stride-2 conv / transposed conv shapes: tuned direct kernel vs F(4x4, 3x3) over parity sub-filters / output phases")
    ap.add_argument("--wgrad", action="store_true", help="time the weight-gradient kernels instead (direct slab kernel vs Winograd F(3x3, 2x2))")
    a = ap.parse_args()
    ops.TUNE_COLD = a.cold
    ops.TUNE_ROUNDS = 3
    lib = L.load()
    wid = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs()
    dev = torch.device("cuda:0")
    if a.shapes:
        SHAPES[:] = [tuple(int(v) for v in t.split(",")) for t in a.shapes.split(";")]
    if a.wgrad:
        nw = lib.crdr_conv2d_wgrad_num_configs()
        for ci, co, hw in SHAPES:
            x = torch.randn(a.bs, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            dy = torch.randn(a.bs, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            g = torch.zeros(co, ci, 3, 3, device=dev)
            res = {}
            for name, cfgs in (("direct", range(0 if a.wino_only else nw - 1)), ("winograd", [nw - 1]), ("f3x3_4x4", [nw])):
                best = (1e9, 0, 0)
                for cfg in cfgs:
                    for ls in range(9):
                        try:
                            t = ops._time_call(lambda: ops.conv2d_wgrad_raw(dy, x, g, (3, 3), 1, 1, False, algo=(cfg + 1) | (ls << 8), defer=False), reps=3)
                        except (L.CrdrHipError, AssertionError):
                            continue
                        best = min(best, (t, cfg, ls))
                res[name] = best
            fl = 2.0 * a.bs * hw * hw * ci * co * 9
            d, w, w4 = res["direct"], res["winograd"], res["f3x3_4x4"]
            print(f"wgrad {co:4d}x{ci:4d} @{hw:3d}: direct {d[0] * 1e3:8.1f} us ({fl / d[0] / 1e9:6.1f} TF, cfg {d[1]} split {1 << d[2]})   winograd {w[0] * 1e3:8.1f} us "
                  f"({fl / w[0] / 1e9:6.1f} TF-eq, split {1 << w[2]})   x{d[0] / w[0]:.2f}   F(3x3,4x4) {w4[0] * 1e3:8.1f} us ({fl / w4[0] / 1e9:6.1f} TF-eq, split {1 << w4[2]}) "
                  f"x{d[0] / w4[0]:.2f}", flush=True)
        return
    if a.k5s2:
        for tr, ci, co, hw in [(1, 256, 256, 64), (0, 192, 192, 128), (0, 256, 256, 128), (1, 192, 192, 64), (1, 256, 256, 32), (0, 192, 192, 64), (0, 256, 256, 64), (1, 192, 192, 32)]:
            x = torch.randn(a.bs, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            wt = (torch.randn(ci, co, 5, 5, device=dev) if tr else torch.randn(co, ci, 5, 5, device=dev)) * (ci * 25) ** -0.5
            b = torch.randn(co, device=dev)
            wp = ops.pack_weight(wt, transpose=bool(tr))
            out = (2 * hw, 2 * hw) if tr else (hw // 2, hw // 2)
            best = (1e9, 0)
            for cfg in range(lib.crdr_conv2d_num_configs()):
                try:
                    t = ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (5, 5), 2, 2, bool(tr), out, bias=b, flags=1, algo=cfg + 1), reps=3)
                except L.CrdrHipError:
                    continue
                best = min(best, (t, cfg))
            y0 = ops.conv2d_raw(x, wp, co, (5, 5), 2, 2, bool(tr), out, bias=b, flags=1, algo=best[1] + 1)
            y4 = ops.conv2d_raw(x, wp, co, (5, 5), 2, 2, bool(tr), out, bias=b, flags=1, algo=wid + 2)
            err = float((y4 - y0).abs().max() / y0.abs().max())
            t4 = ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (5, 5), 2, 2, bool(tr), out, bias=b, flags=1, algo=wid + 2), reps=3)
            fl = 2.0 * a.bs * (hw * hw if tr else out[0] * out[1]) * ci * co * 25
            print(f"{'T' if tr else 'C'} {ci:4d}->{co:4d} k5s2 in{hw:3d}: direct {best[0] * 1e3:8.1f} us ({fl / best[0] / 1e9:6.1f} TF, cfg {best[1]})   F(4x4) {t4 * 1e3:8.1f} us "
                  f"({fl / t4 / 1e9:6.1f} TF-eq) x{best[0] / t4:.2f}   max rel diff {err:.2e}", flush=True)
        return
    kk, shapes = (5, [(320, 4256, 16), (320, 2016, 16), (32, 4032, 16), (32, 2240, 16), (224, 128, 16), (320, 224, 16), (128, 224, 16)]) if a.k5 else (3, SHAPES)
    for ci, co, hw in shapes:
        x = torch.randn(a.bs, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(co, ci, kk, kk, device=dev) * (ci * kk * kk) ** -0.5
        b = torch.randn(co, device=dev)
        wp = ops.pack_weight(wt, transpose=False)
        best = (1e9, 0)
        for cfg in range(0 if a.wino_only else lib.crdr_conv2d_num_configs()):
            try:
                t = min(ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (kk, kk), 1, kk // 2, False, (hw, hw), bias=b, flags=3, algo=(cfg + 1) | (ls << 8)), reps=3)
                        for ls in (range(3) if a.k5 else range(1)))
            except L.CrdrHipError:
                continue
            best = min(best, (t, cfg))
        tw = ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (kk, kk), 1, kk // 2, False, (hw, hw), bias=b, flags=3, algo=wid), reps=3)
        fl = 2.0 * a.bs * hw * hw * ci * co * kk * kk
        t4 = None
        if lib.crdr_conv2d_num_wino_configs() > 2:
            try:   # F(4x4, 3x3) (wino4.hip)
                t4 = ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (kk, kk), 1, kk // 2, False, (hw, hw), bias=b, flags=3, algo=wid + 2), reps=3)
            except L.CrdrHipError:
                pass
        f4 = f"   F(4x4) {t4 * 1e3:8.1f} us ({fl / t4 / 1e9:6.1f} TF-eq) x{best[0] / t4:.2f}" if t4 else ""
        print(f"{ci:4d}->{co:4d} @{hw:3d}: direct {best[0] * 1e3:8.1f} us ({fl / best[0] / 1e9:6.1f} TF, cfg {best[1]})   winograd {tw * 1e3:8.1f} us "
              f"({fl / tw / 1e9:6.1f} TF-eq)   x{best[0] / tw:.2f}{f4}", flush=True)


if __name__ == "__main__":
    main()
