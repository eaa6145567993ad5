import sys, os, subprocess, torch
sys.path.insert(0, ".")
from crdr_amd.hip import lib as L, ops
lib = L.load()
wid = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs()
dev = torch.device("cuda:0")
torch.manual_seed(0)
outs = {}
for ci, co, hw, k, s, tr in [(128, 128, 128, 3, 1, 0), (96, 96, 64, 3, 1, 0), (256, 128, 64, 3, 1, 0), (192, 192, 128, 5, 2, 0), (256, 256, 32, 5, 2, 1), (320, 224, 16, 5, 1, 0), (96, 100, 37, 3, 1, 0)]:
    x = torch.randn(4, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(ci, co, k, k, device=dev) if tr else torch.randn(co, ci, k, k, device=dev)) * (ci * k * k) ** -0.5
    b = torch.randn(co, device=dev)
    wp = ops.pack_weight(wt, transpose=bool(tr))
    p = k // 2
    out = (2 * hw, 2 * hw) if tr else ((hw + s - 1) // s, (hw + s - 1) // s)
    for fl in (0, 1, 3, 5):
        y = ops.conv2d_raw(x, wp, co, (k, k), s, p, bool(tr), out, bias=b, flags=fl, algo=wid + 2)
        outs[(ci, co, hw, k, s, tr, fl)] = y.float().cpu().clone()
torch.save(outs, sys.argv[1])
print("saved", len(outs))
