"""Run one weight-gradient shape a few times (for rocprofv3 --pmc). Usage: python3 tools/pmc_wgrad.py CIN H COUT K STRIDE [BS] [ALGO]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.hip import ops  # noqa: E402

ci, h, co, k, s = [int(v) for v in sys.argv[1:6]]
bs = int(sys.argv[6]) if len(sys.argv) > 6 else 16
algo = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
p = k // 2
oh = ops.conv_out_size(h, k, s, p, False)
x = torch.randn(bs, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(bs, co, oh, oh, device=dev).contiguous(memory_format=torch.channels_last)
g = torch.empty(co, ci, k, k, device=dev)
for _ in range(10):
    ops.conv2d_wgrad_raw(dy, x, g, (k, k), s, p, False, algo=algo)
torch.cuda.synchronize()
print("done")
