"""The 8-wave streaming 1x1 variants (forced ids) on the bottleneck shapes of the step: python tools/experiments/stream1x1_probe.py
(run once per library build: CRDR_HIP_LIB=_exp/<name>/libcrdr_hip.so selects a variant build)."""
import sys

import torch

sys.path.insert(0, ".")
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda:0")
    base = lib.crdr_conv2d_num_configs() + 1
    bs = 16
    tot = {}
    for ci, co, hw, res in [(256, 128, 128, 0), (128, 256, 128, 1), (192, 96, 128, 0), (96, 192, 128, 1), (256, 128, 64, 0), (128, 256, 64, 1),
                            (192, 96, 64, 0), (96, 192, 64, 1), (256, 128, 32, 0), (320, 160, 16, 0), (160, 320, 16, 1)]:
        x = torch.randn(bs, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(co, ci, 1, 1, device=dev) * ci ** -0.5
        b = torch.randn(co, device=dev)
        r = torch.randn(bs, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last) if res else None
        wp = ops.pack_weight(wt, transpose=False)
        out = []
        ref = None
        for v in range(lib.crdr_conv2d_num_stream_configs()):
            try:
                fl = (1 | 16) if res else (1 | 2)
                y = ops.conv2d_raw(x, wp, co, (1, 1), 1, 0, False, (hw, hw), bias=b, flags=fl, res=r, algo=base + v)
                t = ops._time_call(lambda: ops.conv2d_raw(x, wp, co, (1, 1), 1, 0, False, (hw, hw), bias=b, flags=fl, res=r, algo=base + v), reps=5)
            except L.CrdrHipError:
                continue
            if ref is None:
                ref = y.clone()
            same = bool((y == ref).all())
            out.append(f"v{v} {t * 1e3:6.1f}us{'' if same else ' DIFF'}")
            tot[v] = tot.get(v, 0) + t
        print(f"{ci:4d}->{co:4d} @{hw:3d}{' +res' if res else '     '}: " + "  ".join(out), flush=True)
    print("sum over shapes (variants that ran everywhere only comparable):", {v: round(t * 1e3, 1) for v, t in tot.items()})


if __name__ == "__main__":
    main()
