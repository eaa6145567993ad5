import os, sys, torch
sys.path.insert(0, "/root/repo")
from crdr_amd.models.layer.elic_layers import ResidualBottleneckBlocks, BetaCondResidualBottleneckBlocks
from crdr_amd.models.layer.cheng_nlam import ChengNLAM
from crdr_amd.hip import functional as HF
torch.manual_seed(0)
dev = torch.device("cuda:0")
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))

def run(m, x, old, affine):
    for p in m.parameters(): p.grad = None
    xd = x.clone().requires_grad_(True)
    s = torch.rand(x.shape[1], device=dev) + 0.5; t = torch.rand(x.shape[1], device=dev)
    s.requires_grad_(True); t.requires_grad_(True)
    aff = (s, t) if affine else None
    if old:
        y = xd
        for i in range(m.num_blocks):
            y = getattr(m, f"block{i}")(y, affine=aff if i == m.num_blocks - 1 else None)
    else:
        y = m(xd, affine=aff)
    g = torch.Generator(device="cpu").manual_seed(1)
    cot = torch.randn(y.shape, generator=g).to(dev)
    (y * cot).sum().backward()
    return y.detach(), xd.grad, {k: p.grad.clone() for k, p in m.named_parameters()}, (s.grad, t.grad)

m = ResidualBottleneckBlocks(192, 96).to(dev)
x = torch.randn(2, 192, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
torch.manual_seed(5)
for affine in (False, True):
    torch.manual_seed(5); yo, gxo, gpo, sto = run(m, x, True, affine)
    torch.manual_seed(5); yn, gxn, gpn, stn = run(m, x, False, affine)
    print("affine", affine, "y", rel(yn, yo), "dx", rel(gxn, gxo))
    for k in gpo:
        e = rel(gpn[k], gpo[k])
        if e > 1e-5: print("   ", k, e)
    if affine: print("   dscale", rel(stn[0], sto[0]), "dshift", rel(stn[1], sto[1]))
# NLAM: chain vs per-branch
def run_nlam(m, x, old):
    for p in m.parameters(): p.grad = None
    xd = x.clone().requires_grad_(True)
    if old:
        trunk = m.trunk_block(xd); attn = m.attention_block(xd)
        y = m.conv(attn, gate=(xd, trunk))
    else:
        y = m(xd)
    g = torch.Generator(device="cpu").manual_seed(1)
    cot = torch.randn(y.shape, generator=g).to(dev)
    (y * cot).sum().backward()
    return y.detach(), xd.grad, {k: p.grad.clone() for k, p in m.named_parameters()}
m = ChengNLAM(192).to(dev)
yo, gxo, gpo = run_nlam(m, x, True); yn, gxn, gpn = run_nlam(m, x, False)
print("nlam y", rel(yn, yo), "dx", rel(gxn, gxo), "max param err", max(rel(gpn[k], gpo[k]) for k in gpo))
# beta-cond stack
m = BetaCondResidualBottleneckBlocks(256, 128, 512).to(dev)
x2 = torch.randn(2, 256, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
cond = torch.randn(1, 512, 1, 1, device=dev)
def run_bc(old):
    for p in m.parameters(): p.grad = None
    xd = x2.clone().requires_grad_(True); cd = cond.clone().requires_grad_(True)
    s = (torch.arange(256, device=dev) % 7 * 0.1 + 0.5).requires_grad_(True); t = (torch.arange(256, device=dev) % 5 * 0.1).requires_grad_(True)
    if old:
        y = xd
        for i in range(3):
            y = getattr(m, f"block{i}")(y, cd, affine=(s, t) if i == 2 else None)
    else:
        y = m(xd, cd, affine=(s, t))
    g = torch.Generator(device="cpu").manual_seed(1)
    cot = torch.randn(y.shape, generator=g).to(dev)
    (y * cot).sum().backward()
    return y.detach(), xd.grad, cd.grad, {k: p.grad.clone() for k, p in m.named_parameters()}, s.grad, t.grad
a = run_bc(True); b = run_bc(False)
print("betacond y", rel(b[0], a[0]), "dx", rel(b[1], a[1]), "dcond", rel(b[2], a[2]), "max param", max(rel(b[3][k], a[3][k]) for k in a[3]), "ds", rel(b[4], a[4]), "dt", rel(b[5], a[5]))
