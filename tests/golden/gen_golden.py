"""Generate golden vectors from the REFERENCE implementation (run once, in the build container only).

    python tests/golden/gen_golden.py          # needs /root/reference; writes tests/golden/*.npz|json

The reference's Python is imported from /root/reference with inert stand-ins for the third-party packages that
are not installed here (compressai, lpips, pytorch_msssim, cv2, wandb, addict, python_log_indenter,
torchvision): the stand-ins only make `import` succeed, none of their behaviour is recorded. What IS recorded is
the output of the reference's own arithmetic (transforms, InterpChAtt, Fourier conditioning, Charm slice
plumbing, discriminator, losses, header/container, YAML merging) on weights from seeded_weights.py.
Nothing from /root/reference is copied into the repository; only inputs->outputs data.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from seeded_weights import fill_module_, seeded_input  # noqa: E402


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Inert(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    class _AttrDict(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

    mod("compressai")
    mod("compressai.entropy_models", EntropyBottleneck=_Inert, GaussianConditional=_Inert)
    mod("compressai.ans", RansDecoder=object, RansEncoder=object)
    mod("compressai.models", get_scale_table=lambda *a, **k: None)
    mod("compressai.models.utils", update_registered_buffers=lambda *a, **k: None)
    mod("compressai.layers", GDN=_Inert)
    mod("lpips", LPIPS=_Inert)
    mod("pytorch_msssim", MS_SSIM=_Inert, ms_ssim=lambda *a, **k: None, ssim=lambda *a, **k: None)
    mod("cv2")
    mod("wandb")
    mod("addict", Dict=_AttrDict)

    class _IndentedLoggerAdapter:
        def __init__(self, logger, *a, **k):
            self._l = logger
            self.logger = logger
        def __getattr__(self, n):
            if n in ("add", "sub", "push", "pop"):
                return lambda *a, **k: self
            return getattr(self._l, n)
    mod("python_log_indenter", IndentedLoggerAdapter=_IndentedLoggerAdapter)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=object, ToTensor=object, Normalize=object, RandomCrop=object,
                        RandomHorizontalFlip=object, Resize=object, InterpolationMode=types.SimpleNamespace(BILINEAR=2, BICUBIC=3))
    tv.transforms.functional = mod("torchvision.transforms.functional")


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import logging
    logging.disable(logging.CRITICAL)
    from src.models.subnet.autoencoder.elic_autoencoder import ElicEncoder, ElicDecoder
    from src.models.subnet.autoencoder.elic_interpca_autoencoder import ElicInterpCaEncoder
    from src.models.subnet.autoencoder.elic_interpca_beta_cond_autoencoder import ElicInterpCaBetaCondDecoder
    from src.models.subnet.hyperprior.minnen20_hyperprior import Minnen20HyperEncoder, Minnen20HyperDecoder
    from src.models.subnet.context_model.minnen20_charm_context_model import Minnen20CharmContextModel
    from src.models.discriminator.module_list_discriminator import ModuleListDiscriminator
    import src.models.discriminator  # registers CLIC21GVAEDiscriminator  # noqa: F401
    from src.models.layer.interp_channel_attention import InterpChAtt
    from src.models.layer.fourier_cond import FourierEmbedding
    from src.losses.distortion_loss import MSELoss
    from src.losses.rate_loss import HificRateLoss, HificVariableRateLoss
    from src.losses.gan_loss import VanillaGANLoss
    from src.utils.codec_utils import HeaderHandler, MultiRateHeaderHandler, save_byte_strings
    from src.utils.options import BaseConfig
    from oracle import crdr_oracle as O

    out = {}
    ca = dict(actv="softplus", use_interp=True, use_bias=True)
    N, H = 2, 64
    x = seeded_input("image", (N, 3, H, H))
    out["in.image"] = npy(x)

    # ---- encoders
    enc1 = ElicEncoder(in_ch=3, out_ch=320, main_ch=192, block_mid_ch=96)
    fill_module_(enc1, "encoder.")
    out["enc.stage1.y"] = npy(enc1(x))
    enc = ElicInterpCaEncoder(rate_level=5, in_ch=3, out_ch=320, main_ch=192, block_mid_ch=96, ca_kwargs=ca)
    fill_module_(enc, "encoder.")
    for q in (0.0, 1.5, 4.0):
        out[f"enc.q{q}.y"] = npy(enc(x, q))
    # gradients (input + three parameters) of <y, r> at q = 1.5
    xg = x.clone().requires_grad_(True)
    r = seeded_input("enc.cot", (N, 320, H // 16, H // 16))
    (enc(xg, 1.5) * r).sum().backward()
    out["enc.q1.5.dx"] = npy(xg.grad)
    sdg = dict(enc.named_parameters())
    for k in ("conv1.weight", "block2.block1.conv.2.weight", "attn4.conv.bias", "interp_ca_list.4.weight", "interp_ca_list.8.bias"):
        out[f"enc.q1.5.grad.{k}"] = npy(sdg[k].grad)

    # ---- decoders
    y = seeded_input("latent", (N, 320, 4, 4), scale=3.0)
    out["in.latent"] = npy(y)
    dec1 = ElicDecoder(in_ch=320, out_ch=3, main_ch=256, block_mid_ch=128, pixel_shuffle=False, use_tanh=False)
    fill_module_(dec1, "decoder.")
    out["dec.stage1.x"] = npy(dec1(y))
    dec = ElicInterpCaBetaCondDecoder(rate_level=5, L=10, max_beta=5.12, cond_ch=512, weight_init=True, in_ch=320, out_ch=3,
                                      main_ch=256, block_mid_ch=128, pixel_shuffle=False, use_tanh=False, use_pi=False, ca_kwargs=ca)
    fill_module_(dec, "decoder.")
    for q, b in ((0.0, 0.0), (1.5, 2.56), (4.0, 5.12), (2.25, 3.84)):
        out[f"dec.q{q}.b{b}.x"] = npy(dec(y, q, beta=b))
    yg = y.clone().requires_grad_(True)
    r = seeded_input("dec.cot", (N, 3, H, H))
    dec.zero_grad()
    (dec(yg, 1.5, beta=2.56) * r).sum().backward()
    out["dec.q1.5.b2.56.dy"] = npy(yg.grad)
    sdg = dict(dec.named_parameters())
    for k in ("conv4.weight", "block1.block0.proj_2.weight", "mlp.0.weight", "attn2.conv.weight", "interp_ca_list.0.weight"):
        out[f"dec.q1.5.b2.56.grad.{k}"] = npy(sdg[k].grad)

    # ---- hyperprior
    he, hd = Minnen20HyperEncoder(320, 192), Minnen20HyperDecoder(192, 640)
    fill_module_(he, "hyperencoder.")
    fill_module_(hd, "hyperdecoder.")
    z = he(y)
    out["henc.z"] = npy(z)
    zq = torch.round(z)
    out["hdec.in"] = npy(zq)
    out["hdec.out"] = npy(hd(zq))

    # ---- Charm plumbing: the reference's slice/support/LRP logic with the oracle's entropy function plugged in
    cm = Minnen20CharmContextModel(num_slices=10, bottleneck_y=320, hyper_out_ch=640, max_support_slices=5)
    fill_module_(cm, "context_model.")
    hyper = seeded_input("hyper", (N, 640, 4, 4), scale=2.0)
    hyper[:, 320:] = hyper[:, 320:].abs() + 0.05
    out["in.hyper"] = npy(hyper)

    def entropy_fn(ysl, params, is_train):
        mu, sg = params.chunk(2, 1)
        return O.gaussian_conditional(ysl, mu, sg, None)
    with torch.no_grad():
        yh, lik, qlik = cm(y, hyper, entropy_fn, is_train=False, calc_q_likelihood=True)
    out["charm.y_hat"], out["charm.lik"] = npy(yh), npy(lik)

    # ---- discriminator
    D = ModuleListDiscriminator(_subd_type="CLIC21GVAEDiscriminator", _num_subd=5, in_ch=3, out_ch=1, main_ch=64, norm_type="none")
    fill_module_(D, "")
    for q in (0, 3):
        out[f"disc.q{q}"] = npy(D(x, rate_ind=torch.tensor([q])))

    # ---- small layers
    icl = InterpChAtt(48, 5, **ca)
    fill_module_(icl, "encoder.interp_ca_list.0.")
    xi = seeded_input("ica", (2, 48, 3, 5))
    out["in.ica"] = npy(xi)
    for q in (0.0, 0.25, 1.5, 3.75, 4.0):
        out[f"ica.q{q}"] = npy(icl(xi, q))
    fe = FourierEmbedding(L=10, max_beta=5.12, use_pi=False)
    for b in (0.0, 1.28, 2.56, 3.84, 5.12):
        out[f"fourier.b{b}"] = npy(fe.embed(b))

    # ---- losses
    a, b2 = seeded_input("la", (2, 3, 8, 8)), seeded_input("lb", (2, 3, 8, 8))
    out["in.la"], out["in.lb"] = npy(a), npy(b2)
    out["loss.mse150"] = npy(MSELoss(loss_weight=150)(a, b2))
    bpp, qbpp = torch.tensor([0.31, 0.52]), torch.tensor([0.29, 0.49])
    vr = HificVariableRateLoss(lambda_A=[3.6, 1.8, 0.8, 0.4, 0.1], lambda_B=0.015625, target_rate=[0.08, 0.16, 0.36, 0.72, 1.2])
    out["loss.vrate.q2"] = npy(vr(bpp, qbpp=qbpp, current_iter=1, rate_ind=torch.tensor([2])))   # qbpp > target -> lambda_A
    out["loss.vrate.q3"] = npy(vr(bpp, qbpp=qbpp, current_iter=1, rate_ind=torch.tensor([3])))   # qbpp < target -> lambda_B
    hr = HificRateLoss(lambda_A=0.05, lambda_B=0.015625, target_rate=1.5)
    out["loss.rate.s1"] = npy(hr(bpp, qbpp=qbpp, current_iter=1))
    lg = seeded_input("logit", (2, 1, 4, 4), scale=3.0)
    out["in.logit"] = npy(lg)
    gl = VanillaGANLoss(loss_weight=0.000390625)
    out["loss.gan.g_real"] = npy(gl(lg, is_real=True, is_disc=False))
    out["loss.gan.d_fake"] = npy(gl(lg, is_real=False, is_disc=True))

    np.savez_compressed(os.path.join(HERE, "reference_modules.npz"), **out)

    # ---- container bytes + merged configs (json)
    meta = {}
    yh_hdr = torch.zeros(1, 4, 2, 2)
    yh_hdr[0, 0, 0, 0] = -37.6
    meta["header.single"] = HeaderHandler().encode((512, 768), yh_hdr).hex()
    for q in (0.0, 0.25, 1.5, 4.0):
        meta[f"header.multi.q{q}"] = MultiRateHeaderHandler().encode((512, 768), yh_hdr, rate_ind=q).hex()
    tmp = os.path.join(HERE, "_tmp.bin")
    save_byte_strings(tmp, [b"\x01\x02\x03", b"", b"abcdefgh"])
    meta["container"] = open(tmp, "rb").read().hex()
    os.remove(tmp)
    cfgs = {}
    for name in ("crdr.yaml", "crdr_stage_1.yaml", "crdr_stage_2.yaml", "crdr_stage_3.yaml", "examples/example_1.yaml", "examples/example_2.yaml"):
        d, _, loaded = BaseConfig._file2dict_yaml(os.path.join(REF, "config", name))
        cfgs[name] = {"cfg": d, "n_loaded": len(loaded)}
    meta["configs"] = cfgs
    with open(os.path.join(HERE, "reference_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
