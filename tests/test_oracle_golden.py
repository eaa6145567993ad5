"""Pins the CPU oracle (oracle/crdr_oracle.py) against golden vectors recorded from the reference implementation
(tests/golden/gen_golden.py), and pins the product modules' checkpoint schema at the same time: the weights are
generated per (state-dict key, shape) from the *product* modules' state_dict, so any key or shape that differs
from the reference's changes the numbers."""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden.seeded_weights import seeded_input, seeded_tensor

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_modules.npz"))
META = json.load(open(os.path.join(HERE, "golden", "reference_meta.json")))
CA = dict(actv="softplus", use_interp=True, use_bias=True)


def seeded_sd(module, prefix):
    return {prefix + k: seeded_tensor(prefix + k, v.shape) for k, v in module.state_dict().items()
            if torch.is_floating_point(v) and v.numel() > 0}


def close(got, key, rtol=2e-5, atol=2e-5):
    ref = torch.from_numpy(G[key])
    assert got.shape == ref.shape, (key, got.shape, ref.shape)
    err = (got.detach().float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= atol + rtol * scale, f"{key}: err {err:.3e} scale {scale:.3e}"


@pytest.fixture(scope="module")
def O():
    from oracle import crdr_oracle
    return crdr_oracle


@pytest.fixture(scope="module")
def x():
    t = seeded_input("image", (2, 3, 64, 64))
    assert np.array_equal(t.numpy(), G["in.image"])
    return t


def test_encoder_stage1(O, x):
    from crdr_amd.models.subnet.autoencoder.elic_autoencoder import ElicEncoder
    sd = seeded_sd(ElicEncoder(in_ch=3, out_ch=320, main_ch=192, block_mid_ch=96), "encoder.")
    close(O.encoder(sd, x, None), "enc.stage1.y")


def test_encoder_interpca_and_grads(O, x):
    from crdr_amd.models.subnet.autoencoder.elic_interpca_autoencoder import ElicInterpCaEncoder
    sd = seeded_sd(ElicInterpCaEncoder(rate_level=5, in_ch=3, out_ch=320, main_ch=192, block_mid_ch=96, ca_kwargs=CA), "encoder.")
    for q in (0.0, 1.5, 4.0):
        close(O.encoder(sd, x, q), f"enc.q{q}.y")
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xg = x.clone().requires_grad_(True)
    r = seeded_input("enc.cot", (2, 320, 4, 4))
    (O.encoder(sdg, xg, 1.5) * r).sum().backward()
    close(xg.grad, "enc.q1.5.dx", rtol=1e-4)
    for k in ("conv1.weight", "block2.block1.conv.2.weight", "attn4.conv.bias", "interp_ca_list.4.weight", "interp_ca_list.8.bias"):
        close(sdg["encoder." + k].grad, f"enc.q1.5.grad.{k}", rtol=1e-4)


def _decoder_sd():
    from crdr_amd.models.subnet.autoencoder.elic_interpca_beta_cond_autoencoder import ElicInterpCaBetaCondDecoder
    m = ElicInterpCaBetaCondDecoder(rate_level=5, L=10, max_beta=5.12, cond_ch=512, weight_init=True, in_ch=320, out_ch=3,
                                    main_ch=256, block_mid_ch=128, pixel_shuffle=False, use_tanh=False, use_pi=False, ca_kwargs=CA)
    return seeded_sd(m, "decoder.")


def test_decoder_betacond_and_grads(O):
    sd = _decoder_sd()
    y = seeded_input("latent", (2, 320, 4, 4), scale=3.0)
    for q, b in ((0.0, 0.0), (1.5, 2.56), (4.0, 5.12), (2.25, 3.84)):
        close(O.decoder(sd, y, q, b), f"dec.q{q}.b{b}.x", rtol=5e-5)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yg = y.clone().requires_grad_(True)
    r = seeded_input("dec.cot", (2, 3, 64, 64))
    (O.decoder(sdg, yg, 1.5, 2.56) * r).sum().backward()
    close(yg.grad, "dec.q1.5.b2.56.dy", rtol=1e-4)
    for k in ("conv4.weight", "block1.block0.proj_2.weight", "mlp.0.weight", "attn2.conv.weight", "interp_ca_list.0.weight"):
        close(sdg["decoder." + k].grad, f"dec.q1.5.b2.56.grad.{k}", rtol=1e-4)


def test_decoder_stage1(O):
    from crdr_amd.models.subnet.autoencoder.elic_autoencoder import ElicDecoder
    sd = seeded_sd(ElicDecoder(in_ch=320, out_ch=3, main_ch=256, block_mid_ch=128, pixel_shuffle=False, use_tanh=False), "decoder.")
    close(O.decoder(sd, seeded_input("latent", (2, 320, 4, 4), scale=3.0)), "dec.stage1.x", rtol=5e-5)


def test_hyperprior(O):
    from crdr_amd.models.subnet.hyperprior.minnen20_hyperprior import Minnen20HyperDecoder, Minnen20HyperEncoder
    sd = seeded_sd(Minnen20HyperEncoder(320, 192), "hyperencoder.")
    sd.update(seeded_sd(Minnen20HyperDecoder(192, 640), "hyperdecoder."))
    y = seeded_input("latent", (2, 320, 4, 4), scale=3.0)
    close(O.hyper_encoder(sd, y), "henc.z")
    close(O.hyper_decoder(sd, torch.from_numpy(G["hdec.in"])), "hdec.out")


def test_charm_plumbing(O):
    from crdr_amd.models.subnet.context_model.minnen20_charm_context_model import Minnen20CharmContextModel
    sd = seeded_sd(Minnen20CharmContextModel(num_slices=10, bottleneck_y=320, hyper_out_ch=640, max_support_slices=5), "context_model.")
    y = seeded_input("latent", (2, 320, 4, 4), scale=3.0)
    hyper = torch.from_numpy(G["in.hyper"])
    with torch.no_grad():
        yh, lik, _ = O.charm_forward(sd, y, hyper, None)
    close(yh, "charm.y_hat", rtol=5e-5)
    close(lik, "charm.lik", rtol=1e-4, atol=1e-6)


def test_discriminator(O, x):
    from crdr_amd.models.discriminator import build_discriminator
    D = build_discriminator(dict(type="ModuleListDiscriminator", _subd_type="CLIC21GVAEDiscriminator", _num_subd=5, in_ch=3,
                                 out_ch=1, main_ch=64, norm_type="none"))
    sd = seeded_sd(D, "")
    for q in (0, 3):
        close(O.discriminator(sd, x, q), f"disc.q{q}")


def test_interp_ca_and_fourier(O):
    from crdr_amd.models.layer.fourier_cond import FourierEmbedding
    from crdr_amd.models.layer.interp_channel_attention import InterpChAtt
    sd = seeded_sd(InterpChAtt(48, 5, **CA), "encoder.interp_ca_list.0.")
    xi = torch.from_numpy(G["in.ica"])
    for q in (0.0, 0.25, 1.5, 3.75, 4.0):
        close(O.interp_ca(sd, "encoder.interp_ca_list.0", xi, q), f"ica.q{q}")
    fe = FourierEmbedding(L=10, max_beta=5.12, use_pi=False)
    for b in (0.0, 1.28, 2.56, 3.84, 5.12):
        close(O.fourier_embed(b), f"fourier.b{b}", atol=1e-6)
        close(fe.embed(b), f"fourier.b{b}", atol=1e-6)


def test_losses(O):
    a, b = torch.from_numpy(G["in.la"]), torch.from_numpy(G["in.lb"])
    close(O.mse_loss(a, b, 150.0), "loss.mse150")
    bpp, qbpp = torch.tensor([0.31, 0.52]), torch.tensor([0.29, 0.49])
    lam = [3.6, 1.8, 0.8, 0.4, 0.1]
    tgt = [0.08, 0.16, 0.36, 0.72, 1.2]
    close(O.rate_loss(bpp, qbpp, lam[2], 0.015625, tgt[2]), "loss.vrate.q2")
    close(O.rate_loss(bpp, qbpp, lam[3], 0.015625, tgt[3]), "loss.vrate.q3")
    close(O.rate_loss(bpp, qbpp, 0.05, 0.015625, 1.5), "loss.rate.s1")
    lg = torch.from_numpy(G["in.logit"])
    close(O.gan_loss(lg, True, False, 0.000390625), "loss.gan.g_real", atol=1e-9)
    close(O.gan_loss(lg, False, True, 1.0), "loss.gan.d_fake")


def test_container_bytes(tmp_path):
    from crdr_amd.utils.codec_utils import HeaderHandler, MultiRateHeaderHandler, load_byte_strings, save_byte_strings
    yh = torch.zeros(1, 4, 2, 2)
    yh[0, 0, 0, 0] = -37.6
    assert HeaderHandler().encode((512, 768), yh).hex() == META["header.single"]
    for q in (0.0, 0.25, 1.5, 4.0):
        h = MultiRateHeaderHandler().encode((512, 768), yh, rate_ind=q)
        assert h.hex() == META[f"header.multi.q{q}"]
        d = MultiRateHeaderHandler().decode(h)
        assert d["img_size"] == (512, 768) and d["max_sample"] == 37 and d["rate_ind"] == q
    p = tmp_path / "x.bin"
    strings = [b"\x01\x02\x03", b"", b"abcdefgh"]
    save_byte_strings(str(p), strings)
    assert p.read_bytes().hex() == META["container"]
    assert load_byte_strings(str(p)) == strings


def test_config_merge():
    from crdr_amd.utils.options import BaseConfig
    root = os.path.join(os.path.dirname(HERE), "config")
    for name, ref in META["configs"].items():
        cfg, _, loaded = BaseConfig._file2dict_yaml(os.path.join(root, name))
        assert cfg == ref["cfg"], name
        assert len(loaded) == ref["n_loaded"]


@pytest.mark.parametrize("name", ["plain", "nosn", "cond"])
def test_hific_discriminator_oracle_matches_reference(name):
    """oracle.hific_discriminator (incl. torch's spectral-norm power iteration) against vectors
    recorded from the reference's HiFiC discriminators (tests/golden/gen_golden_hific.py)."""
    import numpy as np
    from oracle import crdr_oracle as O
    from tests.golden.seeded_weights import seeded_input, seeded_tensor
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_hific.npz"))
    keys = [str(k) for k in g[f"{name}.keys"]]
    main_ch, y_ch, lat = 16, 24, 4
    shapes = {}
    cin = 3 + (lat if name == "cond" else 0)
    for i, (ci, co, k) in zip((0, 2, 4, 6, 8), ((cin, main_ch, 4), (main_ch, 2 * main_ch, 4), (2 * main_ch, 4 * main_ch, 4),
                                                 (4 * main_ch, 8 * main_ch, 4), (8 * main_ch, 1, 1))):
        if name == "nosn":
            shapes[f"model.{i}.weight"] = (co, ci, k, k)
        else:
            shapes[f"model.{i}.weight_orig"] = (co, ci, k, k)
            shapes[f"model.{i}.weight_u"] = (co,)
            shapes[f"model.{i}.weight_v"] = (ci * k * k,)
        shapes[f"model.{i}.bias"] = (co,)
    if name == "cond":
        shapes["latent_conv.0.weight"], shapes["latent_conv.0.bias"] = (lat, y_ch, 1, 1), (lat,)
    assert sorted(shapes) == keys, "state-dict schema differs from the reference"
    sd = {k: seeded_tensor(f"hific.{name}." + k, s).requires_grad_(not k.endswith(("_u", "_v"))) for k, s in shapes.items()}
    x = torch.from_numpy(g["in.x"]).requires_grad_(True)
    kw = {"y_hat": torch.from_numpy(g["in.y"])} if name == "cond" else {}
    ev = O.hific_discriminator(sd, x, "", training=False, use_sn=name != "nosn", **kw)
    np.testing.assert_allclose(ev.detach().numpy(), g[f"{name}.eval"], rtol=2e-5, atol=2e-6)
    uv = {}
    tr = O.hific_discriminator(sd, x, "", training=True, use_sn=name != "nosn", uv_out=uv, **kw)
    np.testing.assert_allclose(tr.detach().numpy(), g[f"{name}.train"], rtol=2e-5, atol=2e-6)
    tr.backward(seeded_input(f"hific.{name}.gy", tuple(tr.shape)))
    np.testing.assert_allclose(x.grad.numpy(), g[f"{name}.train.dx"], rtol=2e-4, atol=2e-6)
    for k, v in uv.items():
        np.testing.assert_allclose(v.numpy(), g[f"{name}.after.{k}"], rtol=2e-5, atol=2e-6)
    for gk in [k for k in g.files if k.startswith(f"{name}.grad.")]:
        got = sd[gk[len(name) + 6:]].grad
        got = got[:, :8] if got.numel() > 20000 else got
        np.testing.assert_allclose(got.numpy(), g[gk], rtol=2e-4, atol=2e-6)


def test_psnr_and_uint8_truncation_match_the_reference():
    """tests/golden/reference_metrics.json was recorded from the reference's `calc_psnr` / `torch2npimg`
    (src/utils/img_utils.py:17-42,102-132) by tests/golden/gen_golden_metrics.py."""
    import hashlib
    import json
    from crdr_amd.utils.img_utils import calc_psnr, tensor2img
    from tests.golden.gen_golden_metrics import images
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_metrics.json")) as f:
        gold = json.load(f)
    assert len(gold) >= 4
    for tag, g in gold.items():
        real, fake = images(tag, tuple(g["shape"]), g["noise"])
        img = tensor2img(fake)
        assert list(img.shape) == g["npimg_shape"] and int(img.astype(np.int64).sum()) == g["npimg_sum"]
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == g["npimg_sha256"], tag
        assert calc_psnr(real, fake) == g["psnr"], (tag, calc_psnr(real, fake), g["psnr"])


def test_train_transform_known_answers():
    """RandomCrop(pad_if_needed, reflect) -> flip -> ToTensor -> Normalize restated in the oracle: hand-checkable cases."""
    from oracle.crdr_oracle import train_transform_sample
    img = np.arange(3 * 4 * 3, dtype=np.uint8).reshape(3, 4, 3) * 7   # 3 rows x 4 cols
    # no padding needed: plain 2x2 window at (1, 2), flipped horizontally
    out = train_transform_sample(img, 2, 1, 2, 1)
    ref = img[1:3, 2:4][:, ::-1].astype(np.float32) / 255.0
    assert np.array_equal(out, ((ref - 0.5) / 0.5).transpose(2, 0, 1))
    # crop 5 > both axes: rows padded by 2 on both sides, cols by 1; reflect WITHOUT repeating the edge
    out = train_transform_sample(img, 5, -2, -1, 0)
    rows = [2, 1, 0, 1, 2]          # reflect of 3 rows padded by 2: r2 r1 | r0 r1 r2 | r1 r0 -> first five
    cols = [1, 0, 1, 2, 3]          # reflect of 4 cols padded by 1: c1 | c0 c1 c2 c3 | c2 -> first five
    ref = img[np.ix_(rows, cols)].astype(np.float32) / 255.0
    assert np.array_equal(out, ((ref - 0.5) / 0.5).transpose(2, 0, 1))
    assert out.dtype == np.float32 and out.min() >= -1.0 and out.max() <= 1.0
