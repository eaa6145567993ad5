"""CPU oracle for the CRDR hot path -- TEST INFRASTRUCTURE ONLY.

A functional restatement, in stock fp32/fp64 torch ops on the CPU, of the reference algorithm
(iwa-shi/CRDR @ /root/reference; citations below are file:line in that tree).  Every function takes a flat
`state_dict` (same key schema as the reference's checkpoints) plus inputs, so the same seeded weights can be
pushed through the reference modules (tests/golden/gen_golden.py, run once in the build container), through
this oracle, and through the HIP product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(crdr_amd/) never does.

Pinned against the reference: everything whose arithmetic lives in the reference's own Python -- transforms,
InterpChAtt, Fourier/beta conditioning, Charm slice plumbing, discriminator, losses, header/container
(tests/golden/*.npz, tests/test_oracle_golden.py).
PARITY UNPINNED: the entropy-model arithmetic (GaussianConditional / EntropyBottleneck likelihoods, CDF tables,
rANS) lives in the third-party package compressai==1.2.4 (pyproject.toml:16) which is not installed and not
vendored; it is restated here from that package's published behaviour as exercised by the reference's call
sites (ste_gaussian_conditional.py:20-27, entropy_bottleneck.py:18-30, hyperprior_model.py:120-198,
minnen20_charm_context_model.py:143-240) and checked only by analytic known answers, round trips and an independent
exact-rational CDF construction (tests/test_entropy_parity.py).  The end-to-end codec at the bottom of this file
(compress / decompress: header, z string, y string) composes these pieces exactly as the reference's model classes do;
the HIP path is held to it byte for byte (tests/test_gpu_codec_parity.py).  The same
holds for LPIPS (lpips==0.1.4, perceptual_loss.py:23): architecture restated, weights not available offline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------


def conv(sd: SD, name: str, x, stride=1, pad=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad)


def convT(sd: SD, name: str, x, stride=2, pad=2, out_pad=1):
    return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad, output_padding=out_pad)


# ReLU with the masks of ANOTHER implementation imposed (tests only: the deterministic gate on the gradients upstream of the quantiser; the
# sites are every ReLU of the generator -- analysis, hyper-analysis, hyper-synthesis, context model, synthesis: a flip anywhere on the way back
# from the loss reaches the analysis transform's gradients).
# Two correct fp32 implementations differ by summation-order noise; a pre-activation inside that noise band around zero lands on the other
# side of it, and in the backward pass each such element adds a finite rank-one term to the weight gradients above it
# (tests/test_conditioning.py).  As with the rounding decisions (forced_round), the oracle can adopt the device's decision -- relu(t) := t * mask
# -- but ONLY where |t| is within MASK_WINDOW of the layer's largest pre-activation; disagreements outside the window are counted and fail
# the test (check_imposed), as does adopting more than MASK_FRACTION of the elements.
MASK_WINDOW = 1e-4
MASK_FRACTION = 1e-5
RELU_IMPOSE = None   # {"masks": {conv name: bool tensor}, "report": {...}} while generator_forward(impose=...) runs its analysis transforms


def relu(t, site: str):
    imp = RELU_IMPOSE
    if imp is None or not t.requires_grad or site not in imp["masks"]:
        return F.relu(t)
    m = imp["masks"][site].to(t.device)
    assert m.shape == t.shape, (site, m.shape, t.shape)
    td = t.detach()
    diff = (td > 0) != m
    r = imp["report"]
    r["elements"] = r.get("elements", 0) + t.numel()
    r["sites"] = r.get("sites", 0) + 1
    nflip = int(diff.sum())
    if nflip:
        scale = float(td.abs().max())
        mag = td.abs()[diff]
        r["flipped"] = r.get("flipped", 0) + nflip
        r["outside"] = r.get("outside", 0) + int((mag > MASK_WINDOW * scale).sum())
        r["worst"] = max(r.get("worst", 0.0), float(mag.max()) / scale)
    return t * m.to(t.dtype)


def check_imposed(report) -> None:
    """every imposed site was visited, no adopted mask element outside the window, and the adopted ones are the rare boundary cases"""
    assert report.get("sites", 0) > 0, report
    assert report.get("outside", 0) == 0, report
    assert report.get("flipped", 0) <= max(2, int(MASK_FRACTION * report.get("elements", 0))), report


def bottleneck(sd: SD, p: str, x, cond=None):
    """x + 1x1(relu(3x3(relu(1x1 x)))) -- elic_layers.py:23-36; with cond the beta projections are added after each
    ReLU and after the last 1x1 (elic_interpca_beta_cond_autoencoder.py:56-66)."""
    y = relu(conv(sd, p + ".conv.0", x), p + ".conv.0")   # (with cond the product keeps relu(.) + proj: its mask is taken before the add too)
    if cond is not None:
        y = y + conv(sd, p + ".proj_1", cond)
    y = relu(conv(sd, p + ".conv.2", y, pad=1), p + ".conv.2")
    if cond is not None:
        y = y + conv(sd, p + ".proj_2", cond)
    y = conv(sd, p + ".conv.4", y)
    if cond is not None:
        y = y + conv(sd, p + ".proj_3", cond)
    return x + y


def res_blocks(sd: SD, p: str, x, cond=None, n=3):
    for i in range(n):  # elic_layers.py:38-53 (res_in_res is off in every shipped config)
        x = bottleneck(sd, f"{p}.block{i}", x, cond)
    return x


def nlam_res(sd: SD, p: str, x):
    y = relu(conv(sd, p + ".c1", x), p + ".c1")
    y = relu(conv(sd, p + ".c2", y, pad=1), p + ".c2")
    return conv(sd, p + ".c3", y) + x  # cheng_nlam.py:31-46


def nlam(sd: SD, p: str, x):
    t, a = x, x
    for i in range(3):
        t = nlam_res(sd, f"{p}.trunk_block.{i}", t)
        a = nlam_res(sd, f"{p}.attention_block.{i}", a)
    return x + t * torch.sigmoid(conv(sd, p + ".conv", a))  # cheng_nlam.py:23-29


def interp_ca_vectors(W, B, q: float):
    """interp_channel_attention.py:39-73: lerp raw weights between floor(q) and min(floor(q)+1, L-1), THEN softplus."""
    L = W.shape[0]
    l = math.floor(q)
    r = min(l + 1, L - 1)
    a = r - q
    w = W[l] * a + W[r] * (1 - a)
    b = B[l] * a + B[r] * (1 - a)
    return F.softplus(w).reshape(1, -1, 1, 1), b.reshape(1, -1, 1, 1)


def interp_ca(sd: SD, p: str, x, q: float):
    s, t = interp_ca_vectors(sd[p + ".weight"], sd[p + ".bias"], float(q))
    return s * x + t


ENC_LAYERS = ("conv1", "block1", "conv2", "block2", "attn2", "conv3", "block3", "conv4", "attn4")
DEC_LAYERS = ("attn1", "conv1", "block1", "conv2", "attn2", "block2", "conv3", "block3", "conv4")


def _enc_layer(sd, p, name, x):
    if name.startswith("conv"):
        return conv(sd, f"{p}.{name}", x, stride=2, pad=2)
    if name.startswith("block"):
        return res_blocks(sd, f"{p}.{name}", x)
    return nlam(sd, f"{p}.{name}", x)


def encoder(sd: SD, x, q: Optional[float] = None, p: str = "encoder"):
    """ElicEncoder.forward (elic_autoencoder.py:57-72); with q, InterpChAtt AFTER each of the 9 stages
    (elic_interpca_autoencoder.py:51-56)."""
    for i, name in enumerate(ENC_LAYERS):
        x = _enc_layer(sd, p, name, x)
        if q is not None:
            x = interp_ca(sd, f"{p}.interp_ca_list.{i}", x, q)
    return x


def fourier_embed(beta: float, L: int = 10, max_beta: float = 5.12, use_pi: bool = False):
    freq = torch.pow(torch.tensor([2.0]), torch.arange(L))  # fourier_cond.py:12-37
    if use_pi:
        freq = freq * math.pi
    nb = (torch.tensor([float(beta)]) / max_beta - 0.5) * 2
    return torch.cat([torch.sin(nb * freq), torch.cos(nb * freq)], 0).unsqueeze(0)


def decoder(sd: SD, y_hat, q: Optional[float] = None, beta: Optional[float] = None, p: str = "decoder", max_beta=5.12, L=10):
    """ElicDecoder / ElicInterpCaDecoder / ElicInterpCaBetaCondDecoder.forward
    (elic_autoencoder.py:101-119, elic_interpca_autoencoder.py:90-97, elic_interpca_beta_cond_autoencoder.py:150-162):
    InterpChAtt BEFORE each stage; beta conditioning only inside the bottleneck blocks; use_tanh False."""
    cond = None
    if beta is not None:
        e = fourier_embed(beta, L=L, max_beta=max_beta).to(y_hat.dtype)
        h = F.relu(F.linear(e, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"]))
        cond = F.linear(h, sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"]).reshape(1, -1, 1, 1)
    x = y_hat
    for i, name in enumerate(DEC_LAYERS):
        if q is not None:
            x = interp_ca(sd, f"{p}.interp_ca_list.{i}", x, q)
        if name.startswith("conv"):
            x = convT(sd, f"{p}.{name}", x)
        elif name.startswith("block"):
            x = res_blocks(sd, f"{p}.{name}", x, cond)
        else:
            x = nlam(sd, f"{p}.{name}", x)
    return x


def hyper_encoder(sd: SD, y, p: str = "hyperencoder"):
    x = relu(conv(sd, p + ".conv1", y, pad=1), p + ".conv1")  # minnen20_hyperprior.py:23-27
    x = relu(conv(sd, p + ".conv2", x, stride=2, pad=2), p + ".conv2")
    return conv(sd, p + ".conv3", x, stride=2, pad=2)


def hyper_decoder(sd: SD, z_hat, p: str = "hyperdecoder"):
    outs = []
    for br in ("hd_mu", "hd_std"):  # minnen20_hyperprior.py:38-57
        x = relu(convT(sd, f"{p}.{br}.conv1", z_hat), f"{p}.{br}.conv1")
        x = relu(convT(sd, f"{p}.{br}.conv2", x), f"{p}.{br}.conv2")
        outs.append(convT(sd, f"{p}.{br}.conv3", x, stride=1, pad=1, out_pad=0))
    return torch.cat(outs, 1)


def slice_transform(sd: SD, p: str, x):
    x = relu(conv(sd, p + ".model.0", x, pad=2), p + ".model.0")  # minnen20_charm_context_model.py:26-38
    x = relu(conv(sd, p + ".model.2", x, pad=2), p + ".model.2")
    return conv(sd, p + ".model.4", x, pad=1)


# ------------------------------------------------------------------------------------------------------------
# entropy models (compressai 1.2.4 semantics -- PARITY UNPINNED, see module docstring)
# ------------------------------------------------------------------------------------------------------------

SCALE_BOUND = 0.11
LIKELIHOOD_BOUND = 1e-9


class _LowerBound(torch.autograd.Function):
    """max(x, bound) whose gradient passes where x >= bound or the gradient pushes x up (grad < 0)."""

    @staticmethod
    def forward(ctx, x, bound):
        b = torch.full((1,), float(bound), dtype=x.dtype)
        ctx.save_for_backward(x, b)
        return torch.max(x, b)

    @staticmethod
    def backward(ctx, g):
        x, b = ctx.saved_tensors
        return ((x >= b) | (g < 0)).to(g.dtype) * g, None


def lower_bound(x, bound: float):
    return _LowerBound.apply(x, bound)


def _phi(x):
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def gaussian_likelihood(values, mu, sigma, scale_bound=SCALE_BOUND):
    """P(values | N(mu, max(sigma, bound)) integrated over +-1/2), floored at 1e-9."""
    s = lower_bound(sigma, scale_bound)
    v = torch.abs(values - mu)
    lik = _phi((0.5 - v) / s) - _phi((-0.5 - v) / s)
    return lower_bound(lik, LIKELIHOOD_BOUND)


def ste_round(x):
    return (torch.round(x) - x).detach() + x  # ste_round.py:4-5


# Window around a rounding boundary (k + 1/2) inside which another correct fp32 implementation may decide differently.
# y - mu is O(10) after ~100 stacked fp32 convs (relative noise ~1e-6 per conv under a different summation order), so
# the two implementations' y - mu differ by up to a few 1e-4 in absolute terms; 5e-4 covers that with little slack.
FORCE_TOL = 5e-4


def check_forced(report, numel: int) -> None:
    """Gate used by every parity test that hands rounding decisions to the oracle: none outside the window, and the
    adopted ones must stay the rare boundary cases they are meant to be (<= 0.1 % of the symbols, at least 2 allowed)."""
    assert report.get("mismatch", 0) == 0, report
    assert report.get("adopted", 0) <= max(2, int(1e-3 * numel)), (report, numel)


def forced_round(v, forced=None, report=None):
    """round(v); where v sits within FORCE_TOL of a rounding boundary the decision of another implementation
    (`forced`, integer-valued) is adopted instead -- fp32 summation-order noise may legitimately flip those --
    while everywhere else the two decisions must agree exactly (recorded in `report["mismatch"]`)."""
    q = torch.round(v.detach())
    if forced is None:
        return q
    f = v.detach() - torch.floor(v.detach())
    near = (f - 0.5).abs() < FORCE_TOL
    if report is not None:
        report["mismatch"] = report.get("mismatch", 0) + int(((q != forced) & ~near).sum())
        report["adopted"] = report.get("adopted", 0) + int(((q != forced) & near).sum())
        report["symbols"] = report.get("symbols", 0) + int(q.numel())
    return torch.where(near, forced.to(q.dtype), q)


def gaussian_conditional(y, mu, sigma, noise=None, scale_bound=SCALE_BOUND, forced=None, report=None):
    """SteGaussianMeanScaleConditional.forward (ste_gaussian_conditional.py:20-27).
    noise given (training): likelihood of y + noise, output ste_round(y - mu) + mu.
    noise None (is_train False): likelihood of, and output, round(y - mu) + mu."""
    v = y - mu
    q = forced_round(v, forced, report)
    if noise is not None:
        return (q - v).detach() + v + mu, gaussian_likelihood(y + noise, mu, sigma, scale_bound)
    qm = q + mu
    return qm, gaussian_likelihood(qm, mu, sigma, scale_bound)


EB_FILTERS = (1, 3, 3, 3, 3, 1)


def eb_param_names(p: str) -> List[str]:
    names = []
    for i in range(5):
        names += [f"{p}._matrix{i}", f"{p}._bias{i}"] + ([f"{p}._factor{i}"] if i < 4 else [])
    return names + [f"{p}.quantiles"]


def eb_logits(sd: SD, p: str, x, detach: bool = False):
    """Cumulative logits of the factorised prior; x is [C, 1, n]."""
    h = x
    for i in range(5):
        m, b = sd[f"{p}._matrix{i}"], sd[f"{p}._bias{i}"]
        if detach:
            m, b = m.detach(), b.detach()
        h = torch.matmul(F.softplus(m), h) + b
        if i < 4:
            f = sd[f"{p}._factor{i}"]
            if detach:
                f = f.detach()
            h = h + torch.tanh(f) * torch.tanh(h)
    return h


def eb_likelihood(sd: SD, p: str, v):
    """v: [N, C, H, W] -> likelihood of the same shape."""
    n, c = v.shape[:2]
    flat = v.transpose(0, 1).reshape(c, 1, -1)
    lo, up = eb_logits(sd, p, flat - 0.5), eb_logits(sd, p, flat + 0.5)
    sign = -torch.sign(lo + up).detach()
    lik = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))
    lik = lower_bound(lik, LIKELIHOOD_BOUND)
    return lik.reshape(c, n, *v.shape[2:]).transpose(0, 1)


def entropy_bottleneck(sd: SD, p: str, z, noise=None, forced=None, report=None):
    """SteEntropyBottleneck.forward (entropy_bottleneck.py:23-30): training -> likelihood of z + noise and
    z_hat = ste_round(z - median) + median; eval -> both from round(z - median) + median."""
    med = sd[p + ".quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    if noise is not None:
        v = z - med
        return (forced_round(v, forced, report) - v).detach() + v + med, eb_likelihood(sd, p, z + noise)
    q = forced_round(z - med.detach(), forced, report) + med.detach()
    return q, eb_likelihood(sd, p, q)


def eb_aux_loss(sd: SD, p: str):
    target = math.log(2 / 1e-9 - 1)
    t = torch.tensor([-target, 0.0, target], dtype=sd[p + ".quantiles"].dtype)
    return torch.abs(eb_logits(sd, p, sd[p + ".quantiles"], detach=True) - t).sum()


def bits_per_image(lik):
    return -(torch.log(lik).sum(dim=tuple(range(1, lik.ndim)))) / math.log(2)  # hyperprior_model.py:41-46


# ------------------------------------------------------------------------------------------------------------
# training input transform (src/dataset/data_transform.py:34-39 builds it from torchvision, which is not installed here:
# RandomCrop(size, pad_if_needed=True, padding_mode='reflect') -> RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5).
# torchvision's RandomCrop pads BOTH sides of a too-small axis by the full deficit (F.pad with [deficit, 0] / [0, deficit])
# and pads PIL images through numpy's `reflect` mode -- no edge repeat -- which is what is restated below)
# ------------------------------------------------------------------------------------------------------------
def train_transform_sample(img_u8, size: int, sy0: int, sx0: int, flip: int):
    """One sample: img_u8 [H][W][3] uint8, crop origin (sy0, sx0) in the coordinates of the UNPADDED image (it may be
    negative / reach past the image exactly when that axis was padded), flip 0 / 1 -> float32 [3][size][size] in [-1, 1]."""
    h, w, _ = img_u8.shape
    ph, pw = max(0, size - h), max(0, size - w)
    pad = np.pad(img_u8, ((ph, ph), (pw, pw), (0, 0)), mode="reflect") if (ph or pw) else img_u8
    c = pad[sy0 + ph:sy0 + ph + size, sx0 + pw:sx0 + pw + size]
    if flip:
        c = c[:, ::-1]
    t = c.astype(np.float32) / np.float32(255.0)             # ToTensor
    return ((t - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)  # Normalize(0.5, 0.5)


# ------------------------------------------------------------------------------------------------------------
# GDN / IGDN (compressai.layers.GDN 1.2.4 as called from balle18_autoencoder.py:16-20,37-41 -- PARITY UNPINNED against the
# package itself; the published definition (Balle et al. 2016) and the parametrizer constants are restated)
# ------------------------------------------------------------------------------------------------------------
GDN_BETA_MIN = 1e-6
GDN_REPARAM_OFFSET = 2.0 ** -18


def gdn_init(channels: int, gamma_init: float = 0.1):
    """stored parameters at initialisation: sqrt(max(v + pedestal, pedestal))"""
    ped = GDN_REPARAM_OFFSET ** 2
    beta = torch.sqrt(torch.clamp(torch.ones(channels) + ped, min=ped))
    gamma = torch.sqrt(torch.clamp(gamma_init * torch.eye(channels) + ped, min=ped))
    return beta, gamma


def gdn(sd: SD, p: str, x, inverse: bool = False):
    """y = x / sqrt(beta + gamma . x^2) (or times, inverse) with NonNegativeParametrizer on both parameters."""
    ped = GDN_REPARAM_OFFSET ** 2
    beta = lower_bound(sd[p + ".beta"], (GDN_BETA_MIN + ped) ** 0.5) ** 2 - ped
    gamma = lower_bound(sd[p + ".gamma"], GDN_REPARAM_OFFSET) ** 2 - ped
    c = x.shape[1]
    norm = F.conv2d(x * x, gamma.reshape(c, c, 1, 1), beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


# ------------------------------------------------------------------------------------------------------------
# Charm context model + full generator forward
# ------------------------------------------------------------------------------------------------------------


def rounding_margin(v) -> float:
    """Smallest distance of any element of v from a rounding boundary (k + 1/2): inputs whose margin is tiny
    can legitimately round differently under a different fp32 summation order."""
    f = v.detach() - torch.floor(v.detach())
    return float((f - 0.5).abs().min())


def charm_forward(sd: SD, y, hyper_out, noise=None, p: str = "context_model", num_slices=10, max_support=5, diag=None,
                  forced=None, report=None):
    """Minnen20CharmContextModel.forward (minnen20_charm_context_model.py:88-141).
    Support = the FIRST min(i, 5) decoded slices; mean and LRP transforms see hyper_mean, scale sees hyper_scale;
    the coded symbol excludes the LRP residual, the slice handed on includes it."""
    ys = torch.chunk(y, num_slices, 1)
    h_mu, h_sc = torch.chunk(hyper_out, 2, 1)
    ns = None if noise is None else torch.chunk(noise, num_slices, 1)
    hats, liks, qliks = [], [], []
    for i, ysl in enumerate(ys):
        sup = hats[:max_support]
        ms = torch.cat([h_mu] + sup, 1)
        ss = torch.cat([h_sc] + sup, 1)
        mu = slice_transform(sd, f"{p}.mean_slice_transforms.{i}", ms)
        sg = slice_transform(sd, f"{p}.scale_slice_transforms.{i}", ss)
        fq = None if forced is None else forced[i]
        yh, lik = gaussian_conditional(ysl, mu, sg, None if ns is None else ns[i], forced=fq, report=report)
        if diag is not None:
            diag["margin"] = min(diag.get("margin", 1.0), rounding_margin(ysl - mu))
        liks.append(lik)
        with torch.no_grad():
            qliks.append(gaussian_conditional(ysl, mu, sg, None, forced=fq)[1])
        lrp = slice_transform(sd, f"{p}.lrp_slice_transforms.{i}", torch.cat([ms, yh], 1))
        hats.append(yh + 0.5 * torch.tanh(lrp))
    return torch.cat(hats, 1), torch.cat(liks, 1), torch.cat(qliks, 1)


def generator_forward(sd: SD, x, q: Optional[float], beta: Optional[float], noise_y=None, noise_z=None, is_train=True, diag=None,
                      forced=None, report=None, impose=None):
    # forced = {"z": integer symbols [N,192,h,w], "y": list of 10 integer-symbol tensors} from another implementation
    """{HyperpriorCharmModel, BetaCondInterpCaHyperpriorCharmModel}.forward + get_rate_summary_dict
    (hyperprior_charm_model.py:41-79; beta_cond_interpca_hyperprior_charm_model.py:34-78; hyperprior_model.py:60-85).
    q None -> stage-1 model (no InterpCA, no beta)."""
    # impose = {"masks": {conv name: mask}, "report": {}}: the ReLU masks of another implementation for every ReLU of this pass (see relu())
    global RELU_IMPOSE
    RELU_IMPOSE = impose
    try:
        return _generator_forward(sd, x, q, beta, noise_y, noise_z, is_train, diag, forced, report)
    finally:
        RELU_IMPOSE = None


def _generator_forward(sd: SD, x, q, beta, noise_y, noise_z, is_train, diag, forced, report):
    n, _, H, W = x.shape
    y = encoder(sd, x, q)
    z = hyper_encoder(sd, y)
    fz = None if forced is None else forced["z"]
    fy = None if forced is None else forced["y"]
    z_hat, z_lik = entropy_bottleneck(sd, "entropy_model_z", z, noise_z if is_train else None, forced=fz, report=report)
    hyper = hyper_decoder(sd, z_hat)
    if diag is not None:
        diag["margin"] = min(diag.get("margin", 1.0), rounding_margin(z - sd["entropy_model_z.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)))
    y_hat, y_lik, y_qlik = charm_forward(sd, y, hyper, noise_y if is_train else None, diag=diag, forced=fy, report=report)
    fake = decoder(sd, y_hat, q, beta)
    if not is_train:
        fake = fake.clamp(-1, 1)
    with torch.no_grad():
        z_qlik = entropy_bottleneck(sd, "entropy_model_z", z, None, forced=fz)[1]
    npix = H * W
    bpp = (bits_per_image(y_lik) + bits_per_image(z_lik)) / npix
    qbpp = (bits_per_image(y_qlik) + bits_per_image(z_qlik)) / npix
    return dict(fake_images=fake, y=y, z=z, y_hat=y_hat, z_hat=z_hat, bpp=bpp, qbpp=qbpp, y_likelihood=y_lik,
                z_likelihood=z_lik, hyper_out=hyper)


# ------------------------------------------------------------------------------------------------------------
# discriminator + losses + the training step
# ------------------------------------------------------------------------------------------------------------


def clic21_discriminator(sd: SD, x, p: str):
    """CLIC21GVAEDiscriminator with norm_type none (clic21_gvae_discriminator.py:27-50): conv3 s1, then
    [conv3 s2, conv3 s1] x3, conv3 s2, head conv3; LeakyReLU(0.2) after all but the head."""
    strides = (1, 2, 1, 2, 1, 2, 1, 2)
    for i, s in enumerate(strides):
        x = F.leaky_relu(conv(sd, f"{p}.model.{2 * i}", x, stride=s, pad=1), 0.2)
    return conv(sd, f"{p}.model.16", x, pad=1)


def discriminator(sd: SD, x, rate_ind, p: str = "subD_list"):
    return clic21_discriminator(sd, x, f"{p}.{int(rate_ind)}")  # module_list_discriminator.py:25-30


def spectral_norm_weight(sd: SD, name: str, training: bool, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm as the reference applies it (hific_discriminator.py:10-13; one power iteration per
    training-mode forward, none in eval): returns (weight_orig / sigma, new u, new v); u, v enter sigma as constants."""
    w, u, v = sd[name + ".weight_orig"], sd[name + ".weight_u"], sd[name + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma, u, v


def hific_discriminator(sd: SD, x, p: str = "", training: bool = False, y_hat=None, use_sn: bool = True, uv_out: Optional[dict] = None):
    """HiFiCDiscriminator / HiFiCConditionalDiscriminator (hific_discriminator.py:24-58): 4x4 convs with padding
    ceil(3/2) = 2 and strides 2, 2, 2, 1, LeakyReLU(0.2), 1x1 head; the conditional variant prepends a 1x1 conv +
    LeakyReLU of the detached latent, nearest-upsampled x16, concatenated to the image."""
    def w_of(i):
        name = f"{p}model.{i}"
        if not use_sn:
            return sd[name + ".weight"]
        w, u, v = spectral_norm_weight(sd, name, training)
        if uv_out is not None:
            uv_out[name + ".weight_u"], uv_out[name + ".weight_v"] = u, v
        return w
    if y_hat is not None:
        c = F.leaky_relu(F.conv2d(y_hat.detach(), sd[p + "latent_conv.0.weight"], sd[p + "latent_conv.0.bias"]), 0.2)
        x = torch.cat((x, F.interpolate(c, scale_factor=16, mode="nearest")), 1)
    for i, stride in zip((0, 2, 4, 6), (2, 2, 2, 1)):
        x = F.leaky_relu(F.conv2d(x, w_of(i), sd[f"{p}model.{i}.bias"], stride=stride, padding=2), 0.2)
    return F.conv2d(x, w_of(8), sd[p + "model.8.bias"])


def mse_loss(real, fake, weight=150.0):
    return weight * F.mse_loss((real + 1) / 2, (fake + 1) / 2)  # distortion_loss.py:41-46


def rate_loss(bpp, qbpp, lambda_a: float, lambda_b: float, target: float):
    w = lambda_a if float(qbpp.detach().mean()) > target else lambda_b  # rate_loss.py:102-106,172-176
    return w * bpp.mean()


def gan_loss(logit, is_real: bool, is_disc: bool, weight: float):
    t = torch.ones_like(logit) if is_real else torch.zeros_like(logit)
    l = F.binary_cross_entropy_with_logits(logit, t)
    return l if is_disc else weight * l  # gan_loss.py:28-31


ALEX_CFG = ((3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1))
LPIPS_SHIFT = (-0.030, -0.088, -0.188)
LPIPS_SCALE = (0.458, 0.448, 0.450)


def lpips_alex(sd: SD, x0, x1, p: str = "lpips"):
    """lpips.LPIPS(net='alex') forward on [-1,1] images (perceptual_loss.py:25-30): scaling layer, AlexNet
    features at the 5 ReLUs (max-pool 3/2 before conv2 and conv3), unit-normalise over channels, squared
    difference, 1x1 `lin` weights, spatial mean, sum over layers. Returns [N]."""
    sh = torch.tensor(LPIPS_SHIFT, dtype=x0.dtype).view(1, 3, 1, 1)
    sc = torch.tensor(LPIPS_SCALE, dtype=x0.dtype).view(1, 3, 1, 1)

    def feats(x):
        x = (x - sh) / sc
        out = []
        for i, (_, _, k, s, pd) in enumerate(ALEX_CFG):
            if i in (1, 2):
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(F.conv2d(x, sd[f"{p}.net.{i}.weight"], sd[f"{p}.net.{i}.bias"], stride=s, padding=pd))
            out.append(x)
        return out

    total = 0
    for i, (a, b) in enumerate(zip(feats(x0), feats(x1))):
        na = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = ((na - nb) ** 2 * sd[f"{p}.lin.{i}"].view(1, -1, 1, 1)).sum(1)
        total = total + d.mean((1, 2))
    return total


STAGE3 = dict(lambda_a=(3.4, 1.3, 0.4, 0.12, 0.05), lambda_b=2 ** -6, target=(0.0,) * 5, w_mse=150.0,
              w_lpips=0.390625, w_gan=0.000390625, rate_level=5)
STAGE1 = dict(lambda_a=0.05, lambda_b=2 ** -6, target=1.5, w_mse=150.0, w_lpips=1.0)


def stage3_g_losses(sd_g: SD, sd_d: SD, sd_lpips: SD, real, q: int, beta: float, noise_y, noise_z, hr_noise=None, cfg=STAGE3,
                    forced=None, hr_forced=None, report=None, impose=None):
    """Generator phase of MultirateBetaCondHrrGanRateDistortionTrainer.optimize_parameters
    (multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:19-64). Returns (loss dict, generator outputs)."""
    out = generator_forward(sd_g, real, float(q), beta, noise_y, noise_z, forced=forced, report=report, impose=impose)
    fake = out["fake_images"]
    if q + 1 > cfg["rate_level"] - 1:
        rel = real
    else:
        with torch.no_grad():
            hn_y, hn_z = hr_noise if hr_noise is not None else (noise_y, noise_z)
            rel = generator_forward(sd_g, real, float(q + 1), beta, hn_y, hn_z, forced=hr_forced, report=report)["fake_images"]
    losses = {
        "distortion": mse_loss(real, fake, cfg["w_mse"]),
        "rate": rate_loss(out["bpp"], out["qbpp"], cfg["lambda_a"][q], cfg["lambda_b"], cfg["target"][q]),
        "perceptual": cfg["w_lpips"] * lpips_alex(sd_lpips, real, fake).mean(),
    }
    with torch.no_grad():
        real_d = discriminator(sd_d, rel.detach(), q)
    fake_g = discriminator({k: v.detach() for k, v in sd_d.items()}, fake, q)
    adv = (gan_loss(real_d - fake_g, False, False, cfg["w_gan"]) + gan_loss(fake_g - real_d, True, False, cfg["w_gan"])) / 2
    losses["adv"] = adv
    losses["total"] = losses["distortion"] + losses["rate"] + beta * (losses["perceptual"] + adv)
    return losses, out


def stage3_d_losses(sd_d: SD, real, fake, q: int):
    """Discriminator phase (same file :88-104): two backward calls in the reference, summed here."""
    fake = fake.detach()
    with torch.no_grad():
        fake_d0 = discriminator(sd_d, fake, q)
    real_d = discriminator(sd_d, real, q)
    l_real = gan_loss(real_d - fake_d0, True, True, 1.0) * 0.5
    fake_d = discriminator(sd_d, fake, q)
    l_fake = gan_loss(fake_d - real_d.detach(), False, True, 1.0) * 0.5
    return {"d_real": l_real, "d_fake": l_fake, "d_total": l_real + l_fake}


def stage1_losses(sd_g: SD, sd_lpips: SD, real, noise_y, noise_z, cfg=STAGE1, forced=None, report=None, impose=None):
    """RateDistortionTrainer.optimize_parameters loss assembly (rate_distortion_trainer.py:57-75)."""
    out = generator_forward(sd_g, real, None, None, noise_y, noise_z, forced=forced, report=report, impose=impose)
    losses = {
        "distortion": mse_loss(real, out["fake_images"], cfg["w_mse"]),
        "rate": rate_loss(out["bpp"], out["qbpp"], cfg["lambda_a"], cfg["lambda_b"], cfg["target"]),
        "perceptual": cfg["w_lpips"] * lpips_alex(sd_lpips, real, out["fake_images"]).mean(),
    }
    losses["total"] = losses["distortion"] + losses["rate"] + losses["perceptual"]
    return losses, out


# ------------------------------------------------------------------------------------------------------------
# codec tables + rANS (compressai 1.2.4 -- PARITY UNPINNED)
# ------------------------------------------------------------------------------------------------------------


def get_scale_table(lo=0.11, hi=256.0, levels=64):
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    # C++ std::round: halves round away from zero (numpy's round would go to even)
    cdf = [0] + [int(math.floor(float(np.float32(p) * np.float32(1 << precision)) + 0.5)) for p in pmf]
    total = sum(cdf)
    assert total > 0
    cdf = [((1 << precision) * c) // total for c in cdf]
    cdf = list(np.cumsum(cdf))
    cdf[-1] = 1 << precision
    for i in range(len(cdf) - 1):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(len(cdf) - 1):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            assert best != -1
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return [int(c) for c in cdf]


def _pmf_to_cdf_table(pmf, tail_mass, pmf_length, max_length):
    table = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = list(pmf[i][: pmf_length[i]]) + [float(tail_mass[i])]
        c = pmf_to_quantized_cdf(prob, 16)
        table[i, : len(c)] = c
    return table


def gaussian_cdf_tables(scale_table=None, tail_mass=1e-9):
    """GaussianConditional.update_scale_table -> (_quantized_cdf, _cdf_length, _offset)."""
    from scipy.stats import norm
    st = get_scale_table() if scale_table is None else scale_table
    mult = -norm.ppf(tail_mass / 2)
    center = torch.ceil(st * mult).int()
    length = 2 * center + 1
    mx = int(length.max())
    samples = torch.abs(torch.arange(mx).int() - center[:, None]).float()
    sc = st.unsqueeze(1).float()
    upper, lower = _phi((0.5 - samples) / sc), _phi((-0.5 - samples) / sc)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    table = _pmf_to_cdf_table(pmf.numpy(), tail[:, 0].numpy(), length.numpy(), mx)
    return table, (length + 2).numpy().astype(np.int32), (-center).numpy().astype(np.int32)


def eb_cdf_tables(sd: SD, p: str = "entropy_model_z"):
    """EntropyBottleneck.update -> (_quantized_cdf, _cdf_length, _offset)."""
    qt = sd[p + ".quantiles"].detach()
    med = qt[:, 0, 1]
    minima = torch.clamp(torch.ceil(med - qt[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(qt[:, 0, 2] - med).int(), min=0)
    start = med - minima
    length = maxima + minima + 1
    mx = int(length.max())
    samples = torch.arange(mx)[None, :] + start[:, None, None]
    lo = eb_logits(sd, p, samples - 0.5, detach=True)
    up = eb_logits(sd, p, samples + 0.5, detach=True)
    sign = -torch.sign(lo + up)
    pmf = torch.abs(torch.sigmoid(sign * up) - torch.sigmoid(sign * lo))[:, 0, :]
    tail = (torch.sigmoid(lo[:, 0, :1]) + torch.sigmoid(-up[:, 0, -1:]))[:, 0]
    table = _pmf_to_cdf_table(pmf.detach().numpy(), tail.detach().numpy(), length.numpy(), mx)
    return table, (length + 2).numpy().astype(np.int32), (-minima).numpy().astype(np.int32)


def build_indexes(sigma, scale_table=None, scale_bound=SCALE_BOUND):
    st = get_scale_table() if scale_table is None else scale_table
    s = torch.clamp(sigma, min=scale_bound)
    idx = torch.full(s.shape, len(st) - 1, dtype=torch.int32)
    for v in st[:-1]:
        idx -= (s <= v).int()
    return idx


_RANS_L = 1 << 31
_MASK64 = (1 << 64) - 1


def rans_encode(symbols, indexes, cdfs, cdf_sizes, offsets, precision=16, bypass=4) -> bytes:
    """Pure-python restatement of compressai's RansEncoder.encode_with_indexes (rans64, 32-bit words)."""
    maxb = (1 << bypass) - 1
    ops = []
    for s, ci in zip(symbols, indexes):
        cdf = cdfs[ci]
        mv = int(cdf_sizes[ci]) - 2
        v = int(s) - int(offsets[ci])
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, mv
        elif v >= mv:
            raw, v = 2 * (v - mv), mv
        ops.append((int(cdf[v]), int(cdf[v + 1] - cdf[v]), False))
        if v == mv:
            nb = 0
            while (raw >> (nb * bypass)) != 0:
                nb += 1
            val = nb
            while val >= maxb:
                ops.append((maxb, maxb + 1, True))
                val -= maxb
            ops.append((val, val + 1, True))
            for j in range(nb):
                b = (raw >> (j * bypass)) & maxb
                ops.append((b, b + 1, True))
    x = _RANS_L
    words = []
    for start, rng, byp in reversed(ops):
        if not byp:
            x_max = ((_RANS_L >> precision) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << precision) + (x % rng) + start
        else:
            freq = 1 << (16 - bypass)
            x_max = ((_RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << bypass) | start
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    return np.array(list(reversed(words)), dtype="<u4").tobytes()


class RansDecoder:
    def __init__(self, data: bytes):
        self.w = np.frombuffer(data, dtype="<u4")
        self.x = int(self.w[0]) | (int(self.w[1]) << 32)
        self.pos = 2

    def _word(self):
        v = int(self.w[self.pos]) if self.pos < len(self.w) else 0
        self.pos += 1
        return v

    def _bits(self, n):
        val = self.x & ((1 << n) - 1)
        self.x >>= n
        if self.x < _RANS_L:
            self.x = (self.x << 32) | self._word()
        return val

    def decode(self, indexes, cdfs, cdf_sizes, offsets, precision=16, bypass=4):
        maxb = (1 << bypass) - 1
        out = []
        for ci in indexes:
            cdf, size = cdfs[ci], int(cdf_sizes[ci])
            mv = size - 2
            cum = self.x & ((1 << precision) - 1)
            s = 0
            while s < size and int(cdf[s]) <= cum:
                s += 1
            s -= 1
            start, freq = int(cdf[s]), int(cdf[s + 1] - cdf[s])
            self.x = freq * (self.x >> precision) + (self.x & ((1 << precision) - 1)) - start
            if self.x < _RANS_L:
                self.x = (self.x << 32) | self._word()
            v = s
            if v == mv:
                val = self._bits(bypass)
                nb = val
                while val == maxb:
                    val = self._bits(bypass)
                    nb += val
                raw = 0
                for j in range(nb):
                    raw |= self._bits(bypass) << (j * bypass)
                v = raw >> 1
                v = -v - 1 if raw & 1 else v + mv
            out.append(v + int(offsets[ci]))
        return out


# ------------------------------------------------------------------------------------------------------------
# end-to-end codec: compress / decompress of the multi-rate Charm model, on the tables and the Python coder above
# (interpca_hyperprior_charm_model.py:83-149, beta_cond_interpca_hyperprior_charm_model.py:85-149,
#  minnen20_charm_context_model.py:143-240, hyperprior_model.py:120-136, codec_utils.py:22-125, base_model.py:35-58,146-167)
# ------------------------------------------------------------------------------------------------------------
MODEL_STRIDE = 64   # base_model.py:30; = 2**4 (encoder) * 2**2 (hyper-encoder), hyperprior_model.py:131-136

# Relative window around a scale-table entry inside which another correct fp32 implementation may pick the neighbouring
# CDF index: sigma comes out of ~10 stacked fp32 convs (relative noise ~1e-6 each), and build_indexes compares it with
# the 64 table entries -- a discontinuity exactly like rounding.
INDEX_TOL = 1e-4


def forced_indexes(sigma, forced=None, report=None, scale_table=None):
    """build_indexes(sigma); where sigma (after the 0.11 bound) sits within INDEX_TOL (relative) of a table entry the index
    chosen by another implementation (`forced`) is adopted, everywhere else the two must agree exactly
    (`report["idx_mismatch"]`, `report["idx_adopted"]`)."""
    st = get_scale_table() if scale_table is None else scale_table
    idx = build_indexes(sigma, st)
    if forced is None:
        return idx
    forced = forced.to(idx.dtype).reshape(idx.shape)
    s = torch.clamp(sigma, min=SCALE_BOUND).double()
    d = (torch.log(s).unsqueeze(-1) - torch.log(st.double())).abs().amin(-1)
    near = d < INDEX_TOL
    if report is not None:
        report["idx_mismatch"] = report.get("idx_mismatch", 0) + int(((idx != forced) & ~near).sum())
        report["idx_adopted"] = report.get("idx_adopted", 0) + int(((idx != forced) & near).sum())
        report["indexes"] = report.get("indexes", 0) + int(idx.numel())
    return torch.where(near & ((idx - forced).abs() <= 1), forced, idx)


def check_forced_indexes(report) -> None:
    assert report.get("idx_mismatch", 0) == 0, report
    assert report.get("idx_adopted", 0) <= max(2, int(1e-3 * report.get("indexes", 0))), report


def header_bytes(size: Tuple[int, int], y_hat, rate_ind: Optional[float] = None) -> bytes:
    """HeaderHandler.encode / MultiRateHeaderHandler.encode (codec_utils.py:22-39, 82-106): u16 H, u16 W, u8 floor(max|y_hat|)
    [, u8 int(16 q)], little endian."""
    h, w = size
    assert isinstance(h, int) and isinstance(w, int)
    out = np.array([h, w], dtype=np.uint16).tobytes() + np.array(int(torch.max(torch.abs(y_hat))), dtype=np.uint8).tobytes()
    if rate_ind is not None:
        out += np.array(int(float(rate_ind) * 16), dtype=np.uint8).tobytes()
    return out


def header_parse(b: bytes, multirate: bool = True) -> Dict:
    """{Multi,}RateHeaderHandler.decode (codec_utils.py:41-58, 108-125)."""
    hw = np.frombuffer(b[:4], dtype=np.uint16)
    out = {"img_size": (int(hw[0]), int(hw[1])), "max_sample": int(np.frombuffer(b[4:5], dtype=np.uint8)[0])}
    if multirate:
        out["rate_ind"] = float(np.frombuffer(b[5:6], dtype=np.uint8)[0]) / 16
    return out


def pad_image(x, stride: int = MODEL_STRIDE):
    """BaseModel._pad_image (base_model.py:146-152): reflect, bottom / right only."""
    _, _, H, W = x.shape
    pw, ph = int(np.ceil(W / stride) * stride - W), int(np.ceil(H / stride) * stride - H)
    return x if (ph == 0 and pw == 0) else F.pad(x, (0, pw, 0, ph), mode="reflect")


def codec_tables(sd: SD) -> Dict:
    """HyperpriorModel.codec_setup (hyperprior_model.py:120-124): entropy_model_z.update(force=True) and
    entropy_model_y.update_scale_table(get_scale_table(), force=True), as lists the coder indexes."""
    zt, zl, zo = eb_cdf_tables(sd, "entropy_model_z")
    yt, yl, yo = gaussian_cdf_tables()
    return {"z": (zt.tolist(), zl.tolist(), zo.tolist()), "y": (yt.tolist(), yl.tolist(), yo.tolist())}


def compress(sd: SD, x, q: Optional[float], tables: Optional[Dict] = None, forced: Optional[Dict] = None, report: Optional[Dict] = None) -> Dict:
    """{InterpCa,BetaCondInterpCa}HyperpriorCharmModel.compress (interpca_hyperprior_charm_model.py:83-117): one image
    [1,3,H,W] in [-1,1] -> [header, z string, y string] + y_hat, z_hat, likelihoods, predicted bits.
    forced = {"z": int symbols [1,192,h,w], "y": int symbols [1,320,h,w], "idx": CDF indexes [1,320,h,w]} of another
    implementation, adopted only inside FORCE_TOL / INDEX_TOL (see forced_round / forced_indexes)."""
    n, _, H, W = x.shape
    assert n == 1
    report = {} if report is None else report
    tables = codec_tables(sd) if tables is None else tables
    with torch.no_grad():
        xp = pad_image(x)
        y = encoder(sd, xp, q)
        z = hyper_encoder(sd, y)
        med = sd["entropy_model_z.quantiles"].detach()[:, 0, 1].reshape(1, -1, 1, 1)
        fz = None if forced is None else forced["z"].to(torch.float32).reshape(z.shape)
        # EntropyBottleneck.compress: symbols = round(z - medians), index = channel (entropy_bottleneck.py:28, compressai)
        z_sym = forced_round(z - med, fz, report)
        z_hat = z_sym + med
        z_lik = eb_likelihood(sd, "entropy_model_z", z_hat)
        zc = z.shape[1]
        z_idx = torch.arange(zc).reshape(1, zc, 1, 1).expand_as(z_sym)
        z_str = rans_encode(z_sym.reshape(-1).int().tolist(), z_idx.reshape(-1).tolist(), *tables["z"])
        hyper = hyper_decoder(sd, z_hat)
        # forward_compress (minnen20_charm_context_model.py:143-189): the slice loop with is_train False, then ONE stream over
        # the whole latent in NCHW order with indexes from the concatenated scales and means from the concatenated mus
        ys = torch.chunk(y, 10, 1)
        h_mu, h_sc = torch.chunk(hyper, 2, 1)
        fy = None if forced is None else torch.chunk(forced["y"].to(torch.float32).reshape(y.shape), 10, 1)
        hats, liks, mus, sgs, syms = [], [], [], [], []
        for i, ysl in enumerate(ys):
            sup = hats[:5]
            ms, ss = torch.cat([h_mu] + sup, 1), torch.cat([h_sc] + sup, 1)
            mu = slice_transform(sd, f"context_model.mean_slice_transforms.{i}", ms)
            sg = slice_transform(sd, f"context_model.scale_slice_transforms.{i}", ss)
            s = forced_round(ysl - mu, None if fy is None else fy[i], report)
            yh = s + mu
            liks.append(gaussian_likelihood(yh, mu, sg))
            lrp = slice_transform(sd, f"context_model.lrp_slice_transforms.{i}", torch.cat([ms, yh], 1))
            hats.append(yh + 0.5 * torch.tanh(lrp))
            mus.append(mu); sgs.append(sg); syms.append(s)
        y_hat, y_lik, sigma, y_sym = torch.cat(hats, 1), torch.cat(liks, 1), torch.cat(sgs, 1), torch.cat(syms, 1)
        idx = forced_indexes(sigma, None if forced is None else forced["idx"], report)
        y_str = rans_encode(y_sym.reshape(-1).int().tolist(), idx.reshape(-1).tolist(), *tables["y"])
        header = header_bytes((H, W), y_hat, q)
        y_bit, z_bit = float(bits_per_image(y_lik)), float(bits_per_image(z_lik))
    return {"string_list": [header, z_str, y_str], "y_hat": y_hat, "z_hat": z_hat, "y_likelihood": y_lik, "z_likelihood": z_lik,
            "pred_y_bit": y_bit, "pred_y_bpp": y_bit / (H * W), "pred_z_bit": z_bit, "pred_z_bpp": z_bit / (H * W),
            "y_symbols": y_sym.int(), "z_symbols": z_sym.int(), "indexes": idx, "report": report}


def ideal_code_length_bits(symbols, indexes, tables, bypass: int = 4) -> float:
    """sum of -log2(freq / 2^16) over the coded symbols of one stream (+ the escape nibbles of out-of-table symbols): what an
    ideal arithmetic coder spends on these tables; rANS ends within the 64-bit state flush of it."""
    cdfs, sizes, offs = tables
    total = 0.0
    for s, ci in zip(symbols, indexes):
        cdf, mv = cdfs[ci], int(sizes[ci]) - 2
        v = int(s) - int(offs[ci])
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, mv
        elif v >= mv:
            raw, v = 2 * (v - mv), mv
        total += 16.0 - math.log2(cdf[v + 1] - cdf[v])
        if v == mv:
            nb = 0
            while (raw >> (nb * bypass)) != 0:
                nb += 1
            total += bypass * (nb // ((1 << bypass) - 1) + 1 + nb)
    return total


def decompress(sd: SD, strings: Sequence[bytes], beta: Optional[float] = 0.0, tables: Optional[Dict] = None, forced: Optional[Dict] = None,
               report: Optional[Dict] = None, multirate: bool = True) -> Dict:
    """{InterpCa,BetaCondInterpCa}HyperpriorCharmModel.decompress (interpca_hyperprior_charm_model.py:119-149) with
    forward_decompress (minnen20_charm_context_model.py:192-240): header -> z symbols -> hyper-decoder -> per slice
    {mu, sigma -> indexes -> decode -> dequantise -> LRP} -> synthesis -> crop + clamp.
    forced = {"idx": CDF indexes [1,320,h,w]} of another implementation (INDEX_TOL window only)."""
    assert len(strings) == 3
    report = {} if report is None else report
    tables = codec_tables(sd) if tables is None else tables
    hd = header_parse(strings[0], multirate)
    H, W = hd["img_size"]
    q = hd.get("rate_ind")
    zH, zW = int(np.ceil(H / MODEL_STRIDE)), int(np.ceil(W / MODEL_STRIDE))
    with torch.no_grad():
        med = sd["entropy_model_z.quantiles"].detach()[:, 0, 1].reshape(1, -1, 1, 1)
        zc = med.shape[1]
        z_idx = torch.arange(zc).reshape(1, zc, 1, 1).expand(1, zc, zH, zW)
        z_sym = torch.tensor(RansDecoder(strings[1]).decode(z_idx.reshape(-1).tolist(), *tables["z"]), dtype=torch.float32).reshape(1, zc, zH, zW)
        z_hat = z_sym + med   # EntropyBottleneck.dequantize adds the medians back
        hyper = hyper_decoder(sd, z_hat)
        h_mu, h_sc = torch.chunk(hyper, 2, 1)
        dec = RansDecoder(strings[2])
        fi = None if forced is None else torch.chunk(forced["idx"].reshape(1, 320, 4 * zH, 4 * zW), 10, 1)
        hats, syms = [], []
        for i in range(10):
            sup = hats[:5]
            ms, ss = torch.cat([h_mu] + sup, 1), torch.cat([h_sc] + sup, 1)
            mu = slice_transform(sd, f"context_model.mean_slice_transforms.{i}", ms)
            sg = slice_transform(sd, f"context_model.scale_slice_transforms.{i}", ss)
            idx = forced_indexes(sg, None if fi is None else fi[i], report)
            s = torch.tensor(dec.decode(idx.reshape(-1).tolist(), *tables["y"]), dtype=torch.float32).reshape(sg.shape)
            yh = s + mu
            lrp = slice_transform(sd, f"context_model.lrp_slice_transforms.{i}", torch.cat([ms, yh], 1))
            hats.append(yh + 0.5 * torch.tanh(lrp))
            syms.append(s)
        y_hat = torch.cat(hats, 1)
        fake = decoder(sd, y_hat, q, beta if multirate else None)
        fake = fake[:, :, :H, :W].clamp(-1, 1)
    return {"fake_images": fake, "z_hat": z_hat, "y_hat": y_hat, "y_symbols": torch.cat(syms, 1).int(), "z_symbols": z_sym.int(),
            "rate_ind": q, "report": report}
