"""Experiment: Cin=3 forward convs, tap-major (smallc) pack vs generic pack, per tile config."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.hip import ops, lib as L

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3

dev = torch.device("cuda:0")
for (ci, h, co, k, s) in [(3, 256, 64, 3, 1), (3, 256, 192, 5, 2), (3, 256, 32, 3, 1), (3, 128, 64, 3, 1)]:
    p = k // 2
    oh = ops.conv_out_size(h, k, s, p, False)
    x = torch.randn(16, ci, h, h, device=dev)
    x, _ = ops.nhwc(x)
    w = torch.randn(co, ci, k, k, device=dev) * 0.1
    w4 = torch.zeros(co, 4, k, k, device=dev); w4[:, :3] = w
    wt = ops.pack_weight_tapmajor(w)
    wg = ops.pack_weight(w4, False)
    bias = torch.randn(co, device=dev)
    for name, pk, wl in (("tapmajor", wt, 1), ("generic", wg, 0)):
        res = []
        for c in range(L.load().crdr_conv2d_num_configs()):
            try:
                t = timeit(lambda: ops.conv2d_raw(x, pk, co, (k, k), s, p, False, (oh, oh), bias=bias, flags=1, algo=c + 1, wlayout=wl))
            except L.CrdrHipError:
                continue
            res.append((t, c))
        res.sort()
        print(f"C {ci}->{co} k{k}s{s} @{h}: {name:9s} " + "  ".join(f"cfg{c} {t:.0f}us" for t, c in res[:5]), flush=True)
    y = torch.empty(16, co, oh, oh, device=dev).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: y.fill_(1.0))
    print(f"   fill of the output tensor: {t:.0f} us ({y.numel()*4/1e6:.0f} MB)")
