"""GPU parity of the HIP product modules (through the C ABI) against the CPU oracle on the same seeded weights
and inputs: forward values and gradients of every generator sub-network, the entropy models, the full generator
and the discriminator.  Tolerances: outputs rtol 2e-4 of the tensor scale (fp32 MFMA vs CPU fp32 summation
order through ~60 stacked convs); gradients are compared tensor-by-tensor by relative L2 error <= 2e-3."""
import pytest
import torch

from tests.golden.seeded_weights import seeded_input, seeded_tensor

pytestmark = pytest.mark.gpu
CA = dict(actv="softplus", use_interp=True, use_bias=True)


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def seed_module(module, prefix):
    """Fill the (CPU) module with the seeded weights, return the oracle state dict (CPU, fp32)."""
    out = {}
    with torch.no_grad():
        for k, v in module.named_parameters():  # parameters only: buffers (EB target, LPIPS scaling) keep their values
            t = seeded_tensor(prefix + k, v.shape)
            v.copy_(t)
            out[prefix + k] = t.clone()
    return out


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def close(got, ref, what, rtol=2e-5):
    """max |got - ref| <= rtol x max |ref| (default: SURVEY 8d gate (1) with a factor of two, the same figure tests/test_gpu_conv.py holds the
    kernels to; every call of this file measured <= 3.3e-6 in profiles/r5_parity_margins.json); the measured ratio is recorded
    (tests/parity_margins.py) and, once a measurement is committed, the gate tightens to 3 x it."""
    from tests import parity_margins as PM
    got, ref = torch.as_tensor(got).detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
    got, ref = got.reshape(ref.shape) if got.numel() == ref.numel() else got, ref
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    PM.record("fwd:" + what, what, err / scale)
    tol = PM.tolerance("fwd:" + what, rtol)
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e} (rel {err / scale:.2e} > {tol:.2e})"


def grad_close(got, ref, what, tol=2e-3):
    """One gradient tensor (an input gradient) by relative L2 error, recorded and gated like check_grads."""
    from tests import parity_margins as PM
    e = rel(got, ref)
    PM.record("dgrad:" + what, what, e)
    t = PM.tolerance("dgrad:" + what, tol)
    assert e <= t, f"{what}: relative L2 error {e:.3e} > {t:.3e}"


def check_grads(module, prefix, ref_sd, what, tol=2e-3):
    """Every parameter gradient against the oracle's by relative L2 error, gated per parameter group at 3 x the committed
    measurement (profiles/r3_parity_margins.json) and never looser than `tol`."""
    from tests import parity_margins as PM
    bad = []
    for k, p in module.named_parameters():
        r = ref_sd[prefix + k].grad
        if r is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, f"{what}: {k} has a gradient but the oracle has none"
            continue
        if p.grad is None:  # e.g. quantiles under STE rounding: the oracle's autograd yields exact zeros
            assert r.abs().max().item() == 0, f"{what}: {k} got no gradient but the oracle's is non-zero"
            continue
        e = rel(p.grad, r)
        g = "grad:" + PM.group_of(prefix + k)
        PM.record(g, prefix + k, e)
        t = PM.tolerance(g, tol)
        if e > t:
            bad.append((k, e, t))
    assert not bad, f"{what}: gradient mismatch {bad[:8]} ({len(bad)} tensors)"


def grad_sd(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def test_encoder_fwd_bwd():
    from oracle import crdr_oracle as O
    from crdr_amd.models.subnet.autoencoder.elic_interpca_autoencoder import ElicInterpCaEncoder
    m = ElicInterpCaEncoder(rate_level=5, in_ch=3, out_ch=320, main_ch=192, block_mid_ch=96, ca_kwargs=CA)
    sd = seed_module(m, "encoder.")
    m.to(dev())
    x = seeded_input("image", (2, 3, 64, 64))
    r = seeded_input("enc.cot", (2, 320, 4, 4))
    for q in (0.0, 2.5, 4.0):
        close(m(x.to(dev()), q), O.encoder(sd, x, q), f"encoder q={q}")
    sdg = grad_sd(sd)
    xg = x.clone().requires_grad_(True)
    (O.encoder(sdg, xg, 1.5) * r).sum().backward()
    xd = x.to(dev()).requires_grad_(True)
    (m(xd, 1.5) * r.to(dev())).sum().backward()
    grad_close(xd.grad, xg.grad, "dx")
    check_grads(m, "encoder.", sdg, "encoder")


def test_decoder_fwd_bwd():
    from oracle import crdr_oracle as O
    from crdr_amd.models.subnet.autoencoder.elic_interpca_beta_cond_autoencoder import ElicInterpCaBetaCondDecoder
    m = ElicInterpCaBetaCondDecoder(rate_level=5, L=10, max_beta=5.12, cond_ch=512, weight_init=True, in_ch=320, out_ch=3,
                                    main_ch=256, block_mid_ch=128, pixel_shuffle=False, use_tanh=False, use_pi=False, ca_kwargs=CA)
    sd = seed_module(m, "decoder.")
    m.to(dev())
    y = seeded_input("latent", (2, 320, 4, 4), scale=3.0)
    r = seeded_input("dec.cot", (2, 3, 64, 64))
    for q, b in ((0.0, 0.0), (2.25, 3.84)):
        close(m(y.to(dev()).contiguous(memory_format=torch.channels_last), q, b), O.decoder(sd, y, q, b), f"decoder q={q} b={b}")
    sdg = grad_sd(sd)
    yg = y.clone().requires_grad_(True)
    (O.decoder(sdg, yg, 1.5, 2.56) * r).sum().backward()
    yd = y.to(dev()).requires_grad_(True)
    (m(yd, 1.5, 2.56) * r.to(dev())).sum().backward()
    grad_close(yd.grad, yg.grad, "dy")
    check_grads(m, "decoder.", sdg, "decoder")


def test_gauss_cond_fwd_bwd():
    from oracle import crdr_oracle as O
    from crdr_amd.hip import functional as HF
    n, c, h, w = 3, 32, 6, 5
    y = seeded_input("gc.y", (n, c, h, w), 6.0)
    mu = seeded_input("gc.mu", (n, c, h, w), 4.0)
    sg = seeded_input("gc.sg", (n, c, h, w), 2.0)  # includes values below the 0.11 bound and negatives
    noise = seeded_input("gc.noise", (n, c, h, w), 0.5)
    gb = torch.tensor([0.7, 1.3, 0.2])
    gyh = seeded_input("gc.gyh", (n, c, h, w))
    t = [v.clone().requires_grad_(True) for v in (y, mu, sg)]
    yh, lik = O.gaussian_conditional(t[0], t[1], t[2], noise)
    _, qlik = O.gaussian_conditional(y, mu, sg, None)
    ((O.bits_per_image(lik) * gb).sum() + (yh * gyh).sum()).backward()
    d = [v.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True) for v in (y, mu, sg)]
    yh_d, bn, bq, ln, lq = HF.gauss_cond(d[0], d[1], d[2], noise.to(dev()), 0.11, 1e-9, True)
    close(yh_d, yh, "gc yhat", 1e-6)
    close(ln, lik, "gc lik noisy", 2e-5)
    close(lq, qlik, "gc lik quant", 2e-5)
    close(bn, O.bits_per_image(lik), "gc bits noisy", 1e-5)
    close(bq, O.bits_per_image(qlik), "gc bits quant", 1e-5)
    ((bn * gb.to(dev())).sum() + (yh_d * gyh.to(dev())).sum()).backward()
    for a, b, nm in zip(d, t, ("dy", "dmu", "dsigma")):
        close(a.grad, b.grad, "gc " + nm, 2e-4)
    # analytic known answers: p(y = mu, sigma = 1) = erf(1 / (2 sqrt 2)); sigma below the bound clamps to 0.11
    one = torch.zeros(1, 4, 1, 1, device=dev())
    _, _, _, _, l = HF.gauss_cond(one, one, one + 1.0, None, 0.11, 1e-9, True)
    assert abs(l.flatten()[0].item() - 0.3829249) < 1e-6
    _, _, _, _, la = HF.gauss_cond(one, one, one + 0.01, None, 0.11, 1e-9, True)
    _, _, _, _, lb = HF.gauss_cond(one, one, one + 0.11, None, 0.11, 1e-9, True)
    assert torch.equal(la, lb)


def test_gauss_cond_likelihood_against_float64():
    """The kernels' interval likelihood (erfc = exp(-x^2) erfcx(x), csrc/entropy.hip gc_lik) on the grid the oracle is pinned on
    (tests/test_entropy_parity.py::test_gaussian_likelihood_against_float64_scipy: both sides of the sigma bound, the rounding
    boundary, the tail down to the likelihood floor, |v| up to 1e4) plus a dense random sample: against float64 scipy within the
    oracle's own bounds (3e-6 relative + 3e-7 absolute above 1e-5, 3e-7 absolute below -- the erfcf form holds 2e-7 there), bit
    terms within 1e-5, the floor exact; and against the oracle's fp32 result.  Noisy (given-noise) and quantised paths: the same
    function of |v|."""
    import numpy as np
    from scipy.stats import norm
    from oracle import crdr_oracle as O
    from crdr_amd.hip import functional as HF
    sig = np.concatenate([[-1.0, 0.0, 0.01, 0.05, 0.1099, 0.11, 0.1101], np.geomspace(0.12, 256.0, 40)])
    v = np.concatenate([[0.0, 1e-4, 0.25, 0.4999, 0.5, 0.5001, 1.0, 1.5], np.linspace(2.0, 60.0, 59), [200.0, 1e4]])
    S, V = np.meshgrid(sig, v, indexing="ij")
    rng = np.random.default_rng(5)
    s_r = np.exp(rng.uniform(np.log(0.05), np.log(256.0), 200000))
    v_r = np.abs(rng.standard_normal(200000)) * np.maximum(s_r, 0.11) * 2.5 * rng.uniform(0, 1, 200000)
    S = torch.tensor(np.concatenate([S.ravel(), s_r]), dtype=torch.float32)
    V = torch.tensor(np.concatenate([V.ravel(), v_r]), dtype=torch.float32)
    pad = (-S.numel()) % 4
    S, V = torch.cat([S, S[:pad]]), torch.cat([V, V[:pad]])
    mu = torch.full_like(S, 0.375)
    y = mu + V                                       # the noisy path sees |y + 0 - mu| (zero noise given)
    shp = (1, 4, S.numel() // 4, 1)
    d = [t.reshape(shp).to(dev()).contiguous(memory_format=torch.channels_last) for t in (y, mu, S)]
    _, bn, _, ln, _ = HF.gauss_cond(d[0], d[1], d[2], torch.zeros(shp, device=dev()), 0.11, 1e-9, True)
    lik = ln.reshape(-1).double().cpu().numpy()      # (logical NCHW order = the order of the flat operands)
    yv, muv, sv = y, mu, S
    a = np.abs(yv.double().numpy() - muv.double().numpy())
    s64 = np.maximum(sv.double().numpy(), 0.11)
    ref = np.maximum(norm.sf((a - 0.5) / s64) - norm.sf((a + 0.5) / s64), 1e-9)   # (survival form: exact in the tail)
    big = ref > 1e-5
    err = np.abs(lik - ref)
    assert np.all(err[big] <= 3e-6 * ref[big] + 3e-7), float(np.max(err[big] / ref[big]))
    assert np.all(err[~big] <= 3e-7), float(err[~big].max())
    floor = float(np.float32(1e-9))
    assert np.all(lik >= floor) and np.any(lik == floor)
    bits, bits_ref = -np.log2(lik), -np.log2(ref)
    # (bit terms: the likelihood bound above through d(-log2 l) = dl / (l ln 2), plus v_log_f32's own 1e-5)
    assert np.all(np.abs(bits[big] - bits_ref[big]) <= 1.5 * (3e-6 + 3e-7 / ref[big]) + 1e-5 * np.maximum(1.0, bits_ref[big]) + 1e-5)
    # the tail (a >= 1/2, both erfc arguments positive) is relative: 4e-6 + c (1e-6 + 5e-7 xl^2), c = erfc(xl) / (erfc(xl) - erfc(xh)) the
    # cancellation of the difference (wide sigma: the two values are close) and 5e-7 xl^2 the price of an fp32 ARGUMENT (d ln erfc(x) =
    # -2 x dx; a +- 1/2 and the division by sigma round: the erfcf form of the reference shares both terms, operand for operand)
    tail = (a >= 0.5) & (ref > 1e-9)
    cancel = norm.sf((a[tail] - 0.5) / s64[tail]) / ref[tail]
    xl2 = ((a[tail] - 0.5) / s64[tail]) ** 2 / 2.0
    tail_bound = ref[tail] * (4e-6 + cancel * (1e-6 + 5e-7 * xl2))
    w = int(np.argmax(err[tail] / tail_bound))
    assert np.all(err[tail] <= tail_bound), (f"worst: a {a[tail][w]:.6g} sigma {s64[tail][w]:.6g} ref {ref[tail][w]:.6g} lik {lik[tail][w]:.6g} "
                                             f"rel {err[tail][w] / ref[tail][w]:.3g} cancel {cancel[w]:.4g} xl^2 {xl2[w]:.4g} ratio {err[tail][w] / tail_bound[w]:.3g}")
    # the oracle's fp32 (torch erfc) result on the same operands
    ol = O.gaussian_likelihood(yv, muv, sv).double().numpy()
    assert np.all(np.abs(lik - ol) <= 5e-6 * ol + 4e-7), float(np.max(np.abs(lik - ol)))
    tot = float(bn.sum())
    assert abs(tot - float(np.sum(bits_ref))) <= 2e-6 * float(np.sum(bits_ref))


def test_entropy_bottleneck_fwd_bwd_aux():
    from oracle import crdr_oracle as O
    from crdr_amd.models.subnet.entropy_model.entropy_bottleneck import SteEntropyBottleneck
    m = SteEntropyBottleneck(channels=24)
    sd = seed_module(m, "entropy_model_z.")
    m.to(dev())
    z = seeded_input("eb.z", (3, 24, 4, 4), 5.0)
    noise = seeded_input("eb.noise", (3, 24, 4, 4), 0.5)
    gb = torch.tensor([0.7, 1.3, 0.2])
    gzh = seeded_input("eb.gzh", (3, 24, 4, 4))
    sdg = grad_sd(sd)
    zg = z.clone().requires_grad_(True)
    zh, lik = O.entropy_bottleneck(sdg, "entropy_model_z", zg, noise)
    ((O.bits_per_image(lik) * gb).sum() + (zh * gzh).sum()).backward()
    zd = z.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    zh_d, lik_d, bits_d = m(zd, is_train=True, noise=noise.to(dev()), want_bits=True)
    close(zh_d, zh, "eb zhat", 1e-6)
    close(lik_d, lik, "eb lik", 5e-5)
    close(bits_d, O.bits_per_image(lik), "eb bits", 2e-5)
    ((bits_d * gb.to(dev())).sum() + (zh_d * gzh.to(dev())).sum()).backward()
    close(zd.grad, zg.grad, "eb dz", 5e-4)
    check_grads(m, "entropy_model_z.", sdg, "eb params", tol=2e-3)
    # eval path + aux loss
    zq, qlik = O.entropy_bottleneck(sd, "entropy_model_z", z, None)
    zq_d, qlik_d = m(z.to(dev()), is_train=False)
    close(zq_d, zq, "eb eval zhat", 1e-6)
    close(qlik_d, qlik, "eb eval lik", 5e-5)
    m.zero_grad()
    sdg = grad_sd(sd)
    aux = O.eb_aux_loss(sdg, "entropy_model_z")
    aux.backward()
    aux_d = m.loss()
    aux_d.backward()
    close(aux_d, aux, "eb aux", 2e-5)
    close(m.quantiles.grad, sdg["entropy_model_z.quantiles"].grad, "eb aux dquantiles", 5e-4)


def _full_model(stage3=True):
    from crdr_amd.models import build_comp_model
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "_base_", "model")
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(root, "beta_cond_interp_ca_elic_charm.yaml" if stage3 else "elic_charm.yaml"))
    cfg["device"] = "cuda:0"
    model = build_comp_model(ConfigDict(cfg))
    sd = seed_module(model, "")
    return model.to(dev()), sd


@pytest.mark.parametrize("stage3", [True, False], ids=["stage3", "stage1"])
def test_generator_forward_backward(stage3):
    _generator_forward_backward(stage3)


@pytest.mark.parametrize("stage3", [True, False], ids=["stage3", "stage1"])
def test_generator_forward_backward_winograd(stage3):
    """the same comparison with every 3x3 stride-1 convolution and input gradient on the Winograd F(2x2, 3x3) kernel (csrc/wino.hip;
    in production the autotuner picks it per shape)"""
    from crdr_amd.hip import ops
    ops.PREFER_WINOGRAD = True
    try:
        _generator_forward_backward(stage3)
    finally:
        ops.PREFER_WINOGRAD = False


def _generator_forward_backward(stage3):
    from oracle import crdr_oracle as O
    model, sd = _full_model(stage3)
    x = seeded_input("image", (2, 3, 64, 64))
    ny = seeded_input("noise.y", (2, 320, 4, 4), 0.5)
    nz = seeded_input("noise.z", (2, 192, 1, 1), 0.5)
    q, beta = (2.0, 3.84) if stage3 else (None, None)
    kw = dict(rate_ind=q, beta=beta) if stage3 else {}
    model.context_model.record_symbols = []
    out = model.run_model(x, is_train=True, noise={"y": ny.to(dev()), "z": nz.to(dev())}, **kw)
    # rounding decisions of the device are adopted by the oracle only where y - mu sits within 5e-4 (oracle.FORCE_TOL) of a rounding
    # boundary (fp32 summation order may flip those); everywhere else they must agree exactly
    forced = {"y": [t.cpu() for t in model.context_model.record_symbols],
              "z": torch.round(out["z_hat"].detach().cpu() - sd["entropy_model_z.quantiles"][:, 0, 1].reshape(1, -1, 1, 1))}
    model.context_model.record_symbols = None
    sdg = grad_sd(sd)
    rep = {}
    ref = O.generator_forward(sdg, x, q, beta, ny, nz, forced=forced, report=rep)
    O.check_forced(rep, rep.get("symbols", 0))
    loss_ref = O.mse_loss(x, ref["fake_images"]) + 0.4 * ref["bpp"].mean()
    loss_ref.backward()
    close(out["y_hat"], ref["y_hat"], "y_hat", 3e-4)
    close(out["z_hat"], ref["z_hat"], "z_hat", 1e-6)
    close(out["fake_images"], ref["fake_images"], "fake_images", 5e-4)
    close(out["bpp"], ref["bpp"], "bpp", 1e-4)
    close(out["qbpp"], ref["qbpp"], "qbpp", 1e-4)
    from crdr_amd.hip import functional as HF
    mse = HF.sqdiff_sum(out["real_images"], out["fake_images"]) / (x.numel() * 4.0) * 150.0
    close(mse.reshape(()), O.mse_loss(x, ref["fake_images"]), "mse", 1e-4)
    (mse + 0.4 * out["bpp"].mean()).backward()
    check_grads(model, "", sdg, "generator", tol=5e-3)


def test_discriminator_fwd_bwd():
    from oracle import crdr_oracle as O
    from crdr_amd.models.discriminator import build_discriminator
    D = build_discriminator(dict(type="ModuleListDiscriminator", _subd_type="CLIC21GVAEDiscriminator", _num_subd=5, in_ch=3,
                                 out_ch=1, main_ch=64, norm_type="none"))
    sd = seed_module(D, "")
    D.to(dev())
    x = seeded_input("image", (2, 3, 64, 64))
    r = seeded_input("disc.cot", (2, 1, 4, 4))
    sdg = grad_sd(sd)
    xg = x.clone().requires_grad_(True)
    (O.discriminator(sdg, xg, 3) * r).sum().backward()
    xd = x.to(dev()).requires_grad_(True)
    out = D(xd, rate_ind=torch.tensor([3]))
    close(out, O.discriminator(sd, x, 3), "disc logits")
    (out * r.to(dev())).sum().backward()
    grad_close(xd.grad, xg.grad, "dx")
    check_grads(D, "", sdg, "discriminator")


@pytest.mark.parametrize("name", ["plain", "nosn", "cond"])
def test_hific_discriminator_fwd_bwd(name):
    """HiFiC discriminators (spectral norm with its power iteration on the device) against the oracle,
    which is itself pinned to vectors from the reference (tests/test_oracle_golden.py)."""
    from oracle import crdr_oracle as O
    from crdr_amd.models.discriminator.hific_discriminator import HiFiCConditionalDiscriminator, HiFiCDiscriminator
    torch.manual_seed(0)
    if name == "cond":
        D = HiFiCConditionalDiscriminator(in_ch=3, out_ch=1, main_ch=16, y_ch=24, latent_nc=4, use_sn=True)
    else:
        D = HiFiCDiscriminator(in_ch=3, out_ch=1, main_ch=16, use_sn=name != "nosn")
    from tests.golden.seeded_weights import fill_module_
    sd = fill_module_(D, f"hific.{name}.")
    D.to(dev())
    x = seeded_input("hific.x", (2, 3, 32, 48))
    y = seeded_input("hific.y", (2, 24, 2, 3))
    pre = f"hific.{name}."
    osd = {k: v.clone().requires_grad_(not k.endswith(("_u", "_v"))) for k, v in sd.items()}
    kw_o = {"y_hat": y} if name == "cond" else {}
    kw_g = {"y_hat": y.to(dev())} if name == "cond" else {}
    D.eval()
    close(D(x.to(dev()), **kw_g), O.hific_discriminator(osd, x, pre, training=False, use_sn=name != "nosn", **kw_o), "hific eval", 2e-5)
    D.train()
    xo = x.clone().requires_grad_(True)
    uv = {}
    ref = O.hific_discriminator(osd, xo, pre, training=True, use_sn=name != "nosn", uv_out=uv, **kw_o)
    gy = seeded_input(f"hific.{name}.gy", tuple(ref.shape))
    ref.backward(gy)
    xg = x.to(dev()).requires_grad_(True)
    out = D(xg, **kw_g)
    out.backward(gy.to(dev()))
    close(out, ref, "hific train", 2e-5)
    close(xg.grad, xo.grad, "hific dx", 2e-4)
    got = D.state_dict()
    for k, v in uv.items():
        close(got[k[len(pre):]], v, k, 2e-5)
    check_grads(D, pre, osd, f"hific {name}")
