"""One GDN backward at the Balle18 analysis size under rocprofv3 (per-kernel times of the fused / nine-launch forms)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crdr_amd.models.layer.gdn import GDN  # noqa: E402

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
m = GDN(192).to(dev)
x = torch.randn(16, 192, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = m(x)
g = torch.randn_like(y)
for _ in range(6):
    torch.autograd.grad(y, x, g, retain_graph=True)
torch.cuda.synchronize()
