"""Every tile configuration on the discriminator's stride-2 3x3 input gradients (transposed k3 s2, the slowest direct launches of the step):
python tools/experiments/t_k3s2_probe.py [--bs 16].  Prints per shape the time of every configuration that accepts it, plain epilogue."""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--fwd", action="store_true", help="the forward stride-2 convs instead")
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda:0")
    for c, hw in [(64, 128), (128, 64), (256, 32), (512, 16)]:
        if a.fwd:
            x = torch.randn(a.bs, c, 2 * hw, 2 * hw, device=dev).contiguous(memory_format=torch.channels_last)
            wt = torch.randn(c, c, 3, 3, device=dev) * (c * 9) ** -0.5
            wp = ops.pack_weight(wt, transpose=False)
            out, tr = (hw, hw), False
        else:
            x = torch.randn(a.bs, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
            wt = torch.randn(c, c, 3, 3, device=dev) * (c * 9) ** -0.5
            wp = ops.pack_weight(wt, transpose=True)
            out, tr = (2 * hw, 2 * hw), True
        fl = 2.0 * a.bs * hw * hw * c * c * 9
        res = []
        for cfg in range(lib.crdr_conv2d_num_configs()):
            for ls in range(3):
                try:
                    t = ops._time_call(lambda: ops.conv2d_raw(x, wp, c, (3, 3), 2, 1, tr, out, flags=0, algo=(cfg + 1) | (ls << 8)), reps=5)
                except L.CrdrHipError:
                    continue
                res.append((t, cfg, 1 << ls))
        res.sort()
        print(f"{'C' if a.fwd else 'T'} {c}->{c} k3s2 in{x.shape[2]}: " + "  ".join(f"cfg{r[1]}/s{r[2]} {r[0] * 1e3:.0f}us ({fl / r[0] / 1e9:.0f}TF)" for r in res[:8]), flush=True)


if __name__ == "__main__":
    main()
