"""F(4x4) kernel with K splits on the launches that have fewer tiles than CUs: python tools/bench_wino_split.py"""
import sys

import torch

sys.path.insert(0, ".")
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def main():
    lib = L.load()
    wid = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs() + 2
    dev = torch.device("cuda:0")
    ops.TUNE_ROUNDS = 3
    # (transposed, Cin, Cout, input size, kernel, stride)
    for tr, ci, co, hw, k, st in [(1, 4256, 320, 16, 5, 1), (1, 2016, 320, 16, 5, 1), (1, 4032, 32, 16, 5, 1), (0, 512, 256, 32, 3, 1), (0, 192, 192, 64, 5, 2),
                                  (0, 256, 256, 64, 5, 2), (0, 128, 128, 32, 3, 1), (0, 320, 224, 16, 5, 1), (0, 224, 128, 16, 5, 1)]:
        x = torch.randn(16, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(ci, co, k, k, device=dev) if tr else torch.randn(co, ci, k, k, device=dev)) * (ci * k * k) ** -0.5
        wp = ops.pack_weight(wt, transpose=bool(tr))
        oh = hw * st if tr else hw // st
        call = lambda algo: ops.conv2d_raw(x, wp, co, (k, k), st, k // 2, bool(tr), (oh, oh), algo=algo)
        best = (1e9, 0)
        for cfg in range(lib.crdr_conv2d_num_configs()):
            for ls in range(4):
                try:
                    best = min(best, (ops._time_call(lambda: call((cfg + 1) | (ls << 8)), reps=3), (cfg, ls)))
                except L.CrdrHipError:
                    continue
        fl = 2.0 * 16 * hw * hw * ci * co * k * k / (1 if tr else st * st)
        row = f"{'T' if tr else 'C'} {ci:4d}->{co:4d} k{k}s{st} in{hw:3d}: direct {best[0] * 1e3:8.1f} us ({fl / best[0] / 1e9:6.1f} TF, cfg {best[1]})  F(4x4) by splits:"
        for ns in (1, 2, 3, 4, 6, 8):
            try:
                t = ops._time_call(lambda: call(wid | ((ns - 1) << 8)), reps=3)
                row += f"  {ns}: {t * 1e3:7.1f}"
            except L.CrdrHipError:
                row += f"  {ns}:    --  "
        print(row, flush=True)


if __name__ == "__main__":
    main()
