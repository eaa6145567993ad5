import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_step import _opt
from crdr_amd.trainer import build_trainer
from crdr_amd.hip import ops
piece = int(sys.argv[1]); stage = int(sys.argv[2])
opt = _opt(stage); tr = build_trainer(opt); tr.loss_huge_threshold = float("inf")
x = torch.rand(2, 3, 64, 64, device="cuda:0") * 2 - 1
for it in range(2):
    tr.optimize_parameters(it + 1, {"real_images": x, **({"rate_ind": 1, "beta": 2.0} if stage == 3 else {})})
real = tr._stage_input(x)
cond, key = tr._conditions({"rate_ind": 1, "beta": 2.0} if stage == 3 else {})
s = torch.cuda.Stream(); ops.reserve_workspace(0, s)
YH = torch.randn(2, 320, 4, 4, device='cuda:0').contiguous(memory_format=torch.channels_last).requires_grad_(True)
HY = torch.rand(2, 640, 4, 4, device='cuda:0').contiguous(memory_format=torch.channels_last)
X192 = torch.randn(2, 192, 32, 32, device='cuda:0').contiguous(memory_format=torch.channels_last).requires_grad_(True)
NZ = torch.rand(2, 192, 1, 1, device='cuda:0') - 0.5
Z = torch.randn(2, 192, 1, 1, device='cuda:0').requires_grad_(True)
FLAT = torch.zeros(100000, device='cuda:0'); SRC = torch.randn(192, 58, device='cuda:0')
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
def fwd():
    data = {"real_images": real, **({"rate_ind": 1.0, "beta": 2.0} if stage == 3 else {})}
    return tr.run_comp_model(data)
with torch.cuda.graph(g, stream=s):
    if piece == 0:
        with torch.no_grad():
            y = tr.comp_model.encoder(real) if stage == 1 else tr.comp_model.encoder(real, 1.0)
    elif piece == 1:
        with torch.no_grad():
            out = fwd()
    elif piece == 2:
        out = fwd()
    elif piece == 3:
        real_p, fake, bpp, other = fwd()
        l = tr.distortion_loss(real_p, fake) + bpp.mean()
        l.backward()
    elif piece == 4:
        real_p, fake, bpp, other = fwd()
        l = tr.perceptual_loss(real_p, fake)
        l.backward()
    elif piece == 5:
        tr._seg_generator(real, cond, None, 3)
    elif piece == 6:
        m = tr.comp_model
        y = m.encoder(real) if stage == 1 else m.encoder(real, 1.0)
        y.square().mean().backward()
    elif piece == 7:
        m = tr.comp_model
        yh = torch.randn(2, 320, 4, 4, device="cuda:0").contiguous(memory_format=torch.channels_last).requires_grad_(True) if False else YH
        f = m.decoder(yh) if stage == 1 else m.decoder(yh, 1.0, beta=2.0)
        f.square().mean().backward()
    elif piece == 8:
        m = tr.comp_model
        z = m.hyperencoder(YH)
        zh, lik, bits = m.entropy_model_z(z, is_train=True, want_bits=True)
        h = m.hyperdecoder(zh)
        (h.square().mean() + bits.mean()).backward()
    elif piece == 12:
        m = tr.comp_model
        z = m.hyperencoder(YH)
        zh, lik, bits = m.entropy_model_z(z, is_train=True, noise=NZ, want_bits=True)
        h = m.hyperdecoder(zh)
        (h.square().mean() + bits.mean()).backward()
    elif piece == 13:
        m = tr.comp_model
        zh, lik, bits = m.entropy_model_z(Z, is_train=True, noise=NZ, want_bits=True)
        (zh.square().mean() + bits.mean()).backward()
    elif piece == 14:
        m = tr.comp_model
        with torch.no_grad():
            zh, lik, bits = m.entropy_model_z(Z, is_train=True, noise=NZ, want_bits=True)
    elif piece == 15:
        m = tr.comp_model
        z = m.hyperencoder(YH)
        z.square().mean().backward()
    elif piece == 16:
        m = tr.comp_model
        h = m.hyperdecoder(Z)
        h.square().mean().backward()
    elif piece == 17:
        from crdr_amd.hip import functional as HF
        m = tr.comp_model.entropy_model_z
        zh, lik, bits = HF.entropy_bottleneck(Z, m.packed_params().detach(), m._get_medians().detach().reshape(-1), NZ)
        (zh.square().mean() + bits.mean()).backward()
    elif piece == 18:
        from crdr_amd.hip import functional as HF
        m = tr.comp_model.entropy_model_z
        zh, lik, bits = HF.entropy_bottleneck(Z.detach(), m.packed_params(), m._get_medians().detach().reshape(-1), NZ)
        bits.mean().backward()
    elif piece == 19:
        m = tr.comp_model.entropy_model_z
        pp = m.packed_params()
        pp.square().mean().backward()
    elif piece == 20:
        FLAT[3:3 + 576].view(192, 3, 1).add_(torch.ones(192, 3, 1, device="cuda:0"))
    elif piece == 21:
        FLAT[3:3 + 576].view(192, 3, 1).add_(SRC[:, 0:3].reshape(192, 3, 1))
    elif piece == 22:
        FLAT[4:4 + 576].view(192, 3, 1).add_(SRC[:, 0:3].reshape(192, 3, 1))
    elif piece == 23:
        m = tr.comp_model.entropy_model_z
        (m._matrix0 * 2).sum().backward()
    elif piece == 24:
        m = tr.comp_model.entropy_model_z
        torch.cat([m._matrix0.reshape(192, -1), m._bias0.reshape(192, -1)], 1).square().sum().backward()
    elif piece == 9:
        m = tr.comp_model
        yh, _, _ = m.context_model(YH, HY, m.entropy_model_y, is_train=True, want_lik=False)
        (yh.square().mean() + 0).backward()
    elif piece == 10:
        m = tr.comp_model
        c = m.encoder.conv2
        o = c(X192)
        o.square().mean().backward()
    elif piece == 11:
        m = tr.comp_model
        o = m.encoder.block1(X192)
        o.square().mean().backward()
print("piece", piece, "stage", stage, "captured ok")
g.replay(); torch.cuda.synchronize(); print("replayed ok")
