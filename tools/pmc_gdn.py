import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.models.layer.gdn import GDN
dev = torch.device("cuda:0")
m = GDN(192).to(dev)
x = torch.randn(16, 192, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(6):
    m(x)
torch.cuda.synchronize()
