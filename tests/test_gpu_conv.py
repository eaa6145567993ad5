"""GPU parity of the implicit-GEMM conv family (through the C ABI) against stock fp32/fp64 torch on the CPU.

Covers every conv shape family of the CRDR generator / discriminator at reduced spatial size: 1x1, 3x3, 5x5,
stride 2, transposed 5x5 s2 (+output_padding) and 3x3 s1, channel counts that are not powers of two, channel
slices (ld > C), split-K shapes (16x16 spatial, deep K) and every fused epilogue.
Tolerance: fp32 MFMA is an exact fp32 fma chain, so the only difference to the reference is summation order:
|err| <= 2e-5 * sum|a*b| style bound, checked as rtol 2e-4 on the output scale (fp64 reference).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a HIP device"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _close(got, ref, what, rtol=2e-5):   # (exact-fp32 kernels measure ~1e-6 of the output scale against float64: SURVEY 8d gate (1) with a factor of two)
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err/scale:.3e})"


CONV_CASES = [
    # name, N, Cin, H, W, Cout, k, stride, pad
    ("1x1_96_192", 2, 96, 20, 24, 192, 1, 1, 0),
    ("3x3_96_96", 2, 96, 16, 16, 96, 3, 1, 1),
    ("5x5s2_stem", 2, 3, 32, 32, 192, 5, 2, 2),
    ("5x5s2_192_320", 2, 192, 16, 16, 320, 5, 2, 2),
    ("5x5_charm_352_224", 3, 352, 8, 8, 224, 5, 1, 2),
    ("5x5_224_128", 2, 224, 8, 8, 128, 5, 1, 2),
    ("3x3_128_32", 2, 128, 8, 8, 32, 3, 1, 1),
    ("3x3s2_64_64", 2, 64, 18, 22, 64, 3, 2, 1),
    ("3x3_512_1", 2, 512, 4, 4, 1, 3, 1, 1),
    ("3x3_3_64", 1, 3, 24, 24, 64, 3, 1, 1),
    ("1x1_320_160", 1, 320, 6, 10, 160, 1, 1, 0),
    ("11x11s4_alex", 1, 3, 63, 63, 64, 11, 4, 2),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case):
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, k, s, p = case
    dev = _dev()
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, k, k, seed=2, scale=(ci * k * k) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    oh, ow = ref.shape[2:]
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())

    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=False)
    out = ops.conv2d_raw(xd, wp, co, (k, k), s, p, False, (oh, ow), bias=bd, flags=1)
    torch.cuda.synchronize()
    _close(out, ref, name + " fwd")
    if ci <= 4:  # RGB-input convs: tap-major pack, a K-tile = 8 taps x 4 channels
        out2 = ops.conv2d_raw(xd, ops.pack_weight_tapmajor(wd), co, (k, k), s, p, False, (oh, ow), bias=bd, flags=1, wlayout=1)
        _close(out2, ref, name + " fwd (tap-major)")

    wq = ops.pack_weight(wd, transpose=True)  # [T][Cin][Cout]
    dx = ops.conv2d_raw(dyd, wq, ci, (k, k), s, p, True, (h, w))
    _close(dx, xr.grad, name + " dgrad")

    g = torch.zeros_like(wd)
    ops.conv2d_wgrad_raw(dyd, xd, g, (k, k), s, p, accumulate=False)
    _close(g, wr.grad, name + " wgrad")
    ops.conv2d_wgrad_raw(dyd, xd, g, (k, k), s, p, accumulate=True)
    _close(g, 2 * wr.grad, name + " wgrad accumulate")


CONVT_CASES = [
    ("T5x5s2_320_256", 2, 320, 6, 6, 256, 5, 2, 2, 1),
    ("T5x5s2_256_3", 2, 256, 10, 12, 3, 5, 2, 2, 1),
    ("T3x3s1_256_320", 2, 256, 8, 8, 320, 3, 1, 1, 0),
    ("T5x5s2_192_192", 1, 192, 4, 4, 192, 5, 2, 2, 1),
]


@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_convT_fwd_dgrad_wgrad(case):
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, k, s, p, op = case
    dev = _dev()
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(ci, co, k, k, seed=2, scale=(ci * k * k / (s * s)) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, b.double(), stride=s, padding=p, output_padding=op)
    oh, ow = ref.shape[2:]
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())

    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=True)  # rows = Cout, cols = Cin
    out = ops.conv2d_raw(xd, wp, co, (k, k), s, p, True, (oh, ow), bias=bd, flags=1)
    _close(out, ref, name + " fwd")
    wq = ops.pack_weight(wd, transpose=False)  # rows = Cin, cols = Cout
    dx = ops.conv2d_raw(dyd, wq, ci, (k, k), s, p, False, (h, w))
    _close(dx, xr.grad, name + " dgrad")
    g = torch.zeros_like(wd)
    ops.conv2d_wgrad_raw(xd, dyd, g, (k, k), s, p, accumulate=False)
    _close(g, wr.grad, name + " wgrad")


def test_conv_epilogues_and_slices():
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    n, ci, h, w, co = 2, 128, 12, 12, 96
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 3, 3, seed=2, scale=0.03)
    b, v2, sc, sh = _rand(co, seed=3), _rand(co, seed=4), _rand(co, seed=5) + 1.5, _rand(co, seed=6)
    res, gx, gt = _rand(n, co, h, w, seed=7), _rand(n, co, h, w, seed=8), _rand(n, co, h, w, seed=9)
    z = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    view = lambda t: t.double().view(1, -1, 1, 1)
    wp = ops.pack_weight(wt.to(dev), False)
    d = lambda t: t.to(dev)
    chl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)

    out = ops.conv2d_raw(chl(x), wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), vec2=d(v2), flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_VEC2)
    _close(out, F.relu(z) + view(v2), "bias+relu+vec2")
    out = ops.conv2d_raw(chl(x), wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), flags=L.EPI_BIAS | L.EPI_LRELU)
    _close(out, F.leaky_relu(z, 0.2), "bias+lrelu")
    out = ops.conv2d_raw(chl(x), wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), vec2=d(v2), res=chl(res), scale=d(sc), shift=d(sh),
                         flags=L.EPI_BIAS | L.EPI_VEC2 | L.EPI_RES | L.EPI_AFFINE)
    _close(out, (z + view(v2) + res.double()) * view(sc) + view(sh), "bias+vec2+res+affine")
    sig = ops.empty_nhwc(n, co, h, w, dev)
    out = ops.conv2d_raw(chl(x), wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), gate_x=chl(gx), gate_t=chl(gt), sig_out=sig,
                         scale=d(sc), shift=d(sh), flags=L.EPI_BIAS | L.EPI_GATE | L.EPI_AFFINE)
    _close(out, (gx.double() + gt.double() * torch.sigmoid(z)) * view(sc) + view(sh), "gate+affine")
    _close(sig, torch.sigmoid(z), "gate sig")

    # channel slices: read channels [32:160) of a 192-wide tensor, write into channels [64:160) of a 256-wide one
    wide = chl(_rand(n, 192, h, w, seed=10))
    xs = wide[:, 32:160]
    dst = chl(_rand(n, 256, h, w, seed=11))
    before = dst.clone()
    ops.conv2d_raw(xs, wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), flags=L.EPI_BIAS, out=dst[:, 64:160])
    zs = F.conv2d(wide.cpu().double()[:, 32:160], wt.double(), b.double(), padding=1)
    _close(dst[:, 64:160], zs, "slice in/out")
    assert torch.equal(dst[:, :64], before[:, :64]) and torch.equal(dst[:, 160:], before[:, 160:]), "slice write spilled"
    # accumulate
    ops.conv2d_raw(xs, wp, co, (3, 3), 1, 1, False, (h, w), bias=d(b), flags=L.EPI_BIAS | L.EPI_ACCUM, out=dst[:, 64:160])
    _close(dst[:, 64:160], 2 * zs, "accumulate")


def test_epilogue_bwd_narrow_tensors():
    """C <= 4 (the RGB ends): the one-thread-per-row form of the epilogue backward (ebwd_kernel_narrow) against autograd in float64, odd sizes,
    several row blocks"""
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    for c, h, w in ((3, 37, 41), (1, 16, 16), (4, 9, 130)):
        n = 2
        chl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
        z = _rand(n, c, h, w, seed=1).double().requires_grad_(True)
        v2 = _rand(c, seed=2).double().requires_grad_(True)
        sc = (_rand(c, seed=3) + 1.5).double().requires_grad_(True)
        sh = _rand(c, seed=4).double().requires_grad_(True)
        res = _rand(n, c, h, w, seed=5).double().requires_grad_(True)
        dout = _rand(n, c, h, w, seed=6)
        view = lambda t: t.view(1, -1, 1, 1)
        out = F.leaky_relu(z, 0.2)
        out.backward(dout.double())
        dz, _, _, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_LRELU)
        _close(dz, z.grad, f"narrow ebwd lrelu dz c={c}")
        _close(cs[0], z.grad.sum((0, 2, 3)), f"narrow ebwd dbias c={c}")
        z.grad = None
        out = F.relu(z) + view(v2)
        out.backward(dout.double())
        dz, _, _, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_RELU | L.EPI_VEC2, vec2=v2.detach().float().to(dev))
        _close(dz, z.grad, f"narrow ebwd relu dz c={c}")
        _close(cs[0], z.grad.sum((0, 2, 3)), f"narrow ebwd relu dbias c={c}")
        _close(cs[1], v2.grad, f"narrow ebwd dvec2 c={c}")
        for t in (z, v2, sc, sh, res):
            t.grad = None
        out = (z + view(v2) + res) * view(sc) + view(sh)
        out.backward(dout.double())
        dz, gres, _, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_VEC2 | L.EPI_RES | L.EPI_AFFINE,
                                           vec2=v2.detach().float().to(dev), scale=sc.detach().float().to(dev), shift=sh.detach().float().to(dev))
        _close(dz, z.grad, f"narrow ebwd affine dz c={c}")
        _close(gres, res.grad, f"narrow ebwd gres c={c}")
        _close(cs[2], sc.grad, f"narrow ebwd dscale c={c}", rtol=1e-3)
        _close(cs[3], sh.grad, f"narrow ebwd dshift c={c}")


def test_epilogue_bwd():
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    n, c, h, w = 2, 96, 10, 10
    chl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    z = _rand(n, c, h, w, seed=1).double().requires_grad_(True)
    v2 = _rand(c, seed=2).double().requires_grad_(True)
    sc = (_rand(c, seed=3) + 1.5).double().requires_grad_(True)
    sh = _rand(c, seed=4).double().requires_grad_(True)
    res = _rand(n, c, h, w, seed=5).double().requires_grad_(True)
    dout = _rand(n, c, h, w, seed=6)
    view = lambda t: t.view(1, -1, 1, 1)
    # relu + vec2
    out = F.relu(z) + view(v2)
    out.backward(dout.double())
    dz, _, _, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_RELU | L.EPI_VEC2, vec2=v2.detach().float().to(dev))
    _close(dz, z.grad, "ebwd relu dz")
    _close(cs[0], z.grad.sum((0, 2, 3)), "ebwd dbias")
    _close(cs[1], v2.grad, "ebwd dvec2")
    # vec2 + res + affine
    for t in (z, v2, sc, sh, res):
        t.grad = None
    out = (z + view(v2) + res) * view(sc) + view(sh)
    out.backward(dout.double())
    dz, gres, _, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_VEC2 | L.EPI_RES | L.EPI_AFFINE,
                                       vec2=v2.detach().float().to(dev), scale=sc.detach().float().to(dev), shift=sh.detach().float().to(dev))
    _close(dz, z.grad, "ebwd affine dz")
    _close(gres, res.grad, "ebwd gres")
    _close(cs[2], sc.grad, "ebwd dscale", rtol=1e-3)
    _close(cs[3], sh.grad, "ebwd dshift")
    # gate + affine
    gx = _rand(n, c, h, w, seed=7).double().requires_grad_(True)
    gt = _rand(n, c, h, w, seed=8).double().requires_grad_(True)
    for t in (z, sc, sh):
        t.grad = None
    sg = torch.sigmoid(z)
    out = (gx + gt * sg) * view(sc) + view(sh)
    out.backward(dout.double())
    dz, gres, dgt, cs = ops.epilogue_bwd(chl(dout), chl(out.detach().float()), L.EPI_BIAS | L.EPI_GATE | L.EPI_AFFINE,
                                         scale=sc.detach().float().to(dev), shift=sh.detach().float().to(dev),
                                         gate_t=chl(gt.detach().float()), sig=chl(sg.detach().float()))
    _close(dz, z.grad, "ebwd gate dz")
    _close(gres, gx.grad, "ebwd gate gx")
    _close(dgt, gt.grad, "ebwd gate gt")
    _close(cs[2], sc.grad, "ebwd gate dscale", rtol=1e-3)


def test_every_tile_config_and_split():
    """Each tile configuration of the library, with and without split-K / pixel split, on shapes whose sizes are
    not multiples of any tile (forced algorithm ids, the same ids the autotuner uses)."""
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    lib = L.load()
    n, ci, h, w, co, k = 3, 72, 13, 11, 104, 3
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, k, k, seed=2, scale=0.05)
    b = _rand(co, seed=3)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, b.double(), padding=1)
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    wtT = _rand(ci, co, 5, 5, seed=5, scale=0.05)  # transposed 5x5 s2: four phases with different tap counts
    refT = F.conv_transpose2d(x.double(), wtT.double(), b.double(), stride=2, padding=2, output_padding=1)
    xd, dyd, bd = x.to(dev), dy.to(dev), b.to(dev)
    wp, wpT = ops.pack_weight(wt.to(dev), False), ops.pack_weight(wtT.to(dev), True)
    tried = 0
    for c in range(lib.crdr_conv2d_num_configs()):
        for ls in (0, 2):
            algo = (c + 1) | (ls << 8)
            out = ops.conv2d_raw(xd, wp, co, (k, k), 1, 1, False, (h, w), bias=bd, flags=1, algo=algo)
            _close(out, ref, f"fwd cfg {c} split {1 << ls}")
            outT = ops.conv2d_raw(xd, wpT, co, (5, 5), 2, 2, True, tuple(refT.shape[2:]), bias=bd, flags=1, algo=algo)
            _close(outT, refT, f"convT cfg {c} split {1 << ls}")
            tried += 1
    assert tried >= 30
    for c in range(lib.crdr_conv2d_wgrad_num_configs()):
        for ls in (0, 1, 3):
            g = torch.zeros_like(wt, device=dev)
            ops.conv2d_wgrad_raw(dyd, xd, g, (k, k), 1, 1, accumulate=False, algo=(c + 1) | (ls << 8))
            _close(g, wr.grad, f"wgrad cfg {c} split {1 << ls}")


@pytest.mark.parametrize("m,i,o,relu", [(1, 512, 128, False), (1, 20, 512, True), (3, 512, 256, True), (16, 96, 40, False)])
def test_small_linear_path(m, i, o, relu):
    """1x1 conv over <= 16 single-pixel rows takes the GEMV kernels (crdr_linear_fwd / crdr_linear_bwd)."""
    from crdr_amd.hip import functional as HF
    dev = _dev()
    x = _rand(m, i, 1, 1, seed=1)
    wt = _rand(o, i, 1, 1, seed=2, scale=i ** -0.5)
    b = _rand(o, seed=3)
    xr, wr, br = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br)
    ref = ref.relu() if relu else ref
    gy = _rand(*ref.shape, seed=4)
    ref.backward(gy.double())
    spec = HF.ConvSpec(i, o, 1, 1, 0)
    xg = x.to(dev).requires_grad_(True)
    wg = torch.nn.Parameter(wt.to(dev))
    bg = torch.nn.Parameter(b.to(dev))
    out = HF.fused_conv(xg, wg, bg, spec, act="relu" if relu else None)
    assert out.grad_fn is not None and "SmallLinear" in type(out.grad_fn).__name__
    out.backward(gy.to(dev))
    _close(out, ref, "small linear out")
    _close(xg.grad, xr.grad, "small linear dx")
    _close(wg.grad, wr.grad, "small linear dw")
    _close(bg.grad, br.grad, "small linear db")


RGB_CASES = [
    # name, transposed, Cin, Cout, k, stride, pad, out_pad, H
    ("up4_T256_3_k5s2", True, 256, 3, 5, 2, 2, 1, 12),
    ("T64_3_k3s1", True, 64, 3, 3, 1, 1, 0, 10),
    ("D_3_64_k3", False, 3, 64, 3, 1, 1, 0, 20),
    ("stem_3_192_k5s2", False, 3, 192, 5, 2, 2, 0, 22),
    ("alex_3_64_k11s4", False, 3, 64, 11, 4, 2, 0, 63),
]


@pytest.mark.parametrize("case", RGB_CASES, ids=[c[0] for c in RGB_CASES])
def test_rgb_layers_through_fused_conv(case):
    """3-channel ends of the networks: tap-major forward / input-gradient packs, GEMM + col2im for RGB-output transposed
    ops, tap-folded weight gradient -- all behind fused_conv, against fp64 torch."""
    from crdr_amd.hip import functional as HF
    name, tr, ci, co, k, s, p, op, h = case
    dev = _dev()
    x = _rand(2, ci, h, h + 3, seed=1)
    wt = _rand(*((ci, co, k, k) if tr else (co, ci, k, k)), seed=2, scale=(ci * k * k) ** -0.5)
    b = _rand(co, seed=3)
    xr, wr, br = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op) if tr else F.conv2d(xr, wr, br, stride=s, padding=p)
    gy = _rand(*ref.shape, seed=4)
    ref.backward(gy.double())
    spec = HF.ConvSpec(ci, co, k, s, p, transposed=tr, out_pad=op)
    xg = x.to(dev).requires_grad_(True)
    wg, bg = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
    out = HF.fused_conv(xg, wg, bg, spec)
    out.backward(gy.to(dev))
    _close(out, ref, name + " out")
    _close(xg.grad, xr.grad, name + " dx")
    _close(wg.grad, wr.grad, name + " dw")
    _close(bg.grad, br.grad, name + " db")


def test_operands_beyond_2gib(monkeypatch):
    """Forward convs address the input through a buffer descriptor that is re-based per workgroup, so byte offsets past
    2 GiB work (here: a 32-channel slice of a 4096-float pixel stride, 2.6 GB span); the weight-gradient kernel keeps one
    descriptor per operand and ops splits the batch instead."""
    from crdr_amd.hip import ops
    dev = _dev()
    h = w = 400
    base = torch.zeros(1, h, w, 4096, device=dev)
    xs = _rand(1, 32, h, w, seed=1)
    x = base[..., :32].permute(0, 3, 1, 2)
    x.copy_(xs.to(dev))
    assert x.stride(1) == 1 and x.stride(3) == 4096 and h * w * 4096 * 4 > (1 << 31)
    wt = _rand(16, 32, 3, 3, seed=2, scale=0.1)
    b = _rand(16, seed=3)
    out = ops.conv2d_raw(x, ops.pack_weight(wt.to(dev), False), 16, (3, 3), 1, 1, False, (h, w), bias=b.to(dev), flags=1)
    _close(out, F.conv2d(xs.double(), wt.double(), b.double(), padding=1), "conv on a >2 GiB span")
    outT = ops.conv2d_raw(x, ops.pack_weight(_rand(32, 16, 5, 5, seed=4, scale=0.1).to(dev), True), 16, (5, 5), 2, 2, True, (2 * h, 2 * w))
    _close(outT, F.conv_transpose2d(xs.double(), _rand(32, 16, 5, 5, seed=4, scale=0.1).double(), stride=2, padding=2, output_padding=1),
           "convT on a >2 GiB span")
    del base, x, out, outT
    # weight gradient: batch halves
    x4 = _rand(4, 32, 12, 12, seed=1).to(dev).contiguous(memory_format=torch.channels_last)
    dy = _rand(4, 16, 12, 12, seed=3).to(dev).contiguous(memory_format=torch.channels_last)
    gref = ops.conv2d_wgrad_raw(dy, x4, torch.empty(16, 32, 3, 3, device=dev), (3, 3), 1, 1, False)
    monkeypatch.setattr(ops, "_SPAN_LIMIT", 4 * 12 * 12 * 32 * 4 - 1)  # pretend the 4-image batch is too large
    ggot = ops.conv2d_wgrad_raw(dy, x4, torch.empty(16, 32, 3, 3, device=dev), (3, 3), 1, 1, False)
    _close(ggot, gref, "split wgrad", 1e-5)


def test_degenerate_shapes():
    """Empty batch, single pixel, single channel in / out, spatial sizes smaller than the kernel: every entry point
    returns well-formed (possibly empty) results instead of launching on nothing."""
    from crdr_amd.hip import functional as HF
    dev = _dev()
    spec = HF.ConvSpec(8, 12, 3, 1, 1)
    w = torch.nn.Parameter(_rand(12, 8, 3, 3, seed=2).to(dev))
    b = torch.nn.Parameter(_rand(12, seed=3).to(dev))
    empty = torch.zeros(0, 8, 5, 5, device=dev)
    y = HF.fused_conv(empty, w, b, spec, act="relu")
    assert y.shape == (0, 12, 5, 5)
    for h, wd in ((1, 1), (2, 1), (1, 7)):  # smaller than the 3x3 kernel: padding supplies the rest
        x = _rand(2, 8, h, wd, seed=h * 10 + wd)
        xr = x.double().requires_grad_(True)
        ref = F.conv2d(xr, w.detach().cpu().double(), b.detach().cpu().double(), padding=1).relu()
        ref.sum().backward()
        xg = x.to(dev).requires_grad_(True)
        out = HF.fused_conv(xg, w, b, spec, act="relu")
        out.sum().backward()
        _close(out, ref, f"{h}x{wd} out")
        _close(xg.grad, xr.grad, f"{h}x{wd} dx")
    spec1 = HF.ConvSpec(1, 1, 5, 2, 2)  # one channel in and out, strided
    w1 = torch.nn.Parameter(_rand(1, 1, 5, 5, seed=5).to(dev))
    x = _rand(1, 1, 9, 11, seed=6)
    _close(HF.fused_conv(x.to(dev), w1, None, spec1), F.conv2d(x.double(), w1.detach().cpu().double(), stride=2, padding=2), "1->1 k5s2")


STREAM_CASES = [
    # name, N, H, W, Cin, Cout  (M = N*H*W: multiples of 128, ragged, fewer tiles than lanes, more tiles than lanes)
    ("256_128", 2, 32, 32, 256, 128),
    ("96_192_ragged", 3, 7, 9, 96, 192),
    ("192_96", 1, 40, 52, 192, 96),
    ("128_256_many_tiles", 5, 96, 96, 128, 256),
    ("256_100_partial_cols", 2, 17, 23, 256, 100),
    ("64_64", 1, 5, 5, 64, 64),
    ("32_32_one_chunk", 2, 30, 30, 32, 32),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_streaming_1x1_variants(case):
    """The persistent 1x1 kernel (forced ids num_configs + 1 + v) against the fp64 reference and, bit for bit, against the
    tiled kernel -- same MFMA order per output, same epilogue arithmetic -- for every epilogue it implements, including
    channel-slice operands (ld > C), the in-epilogue column sums and accumulation into the output."""
    import ctypes as C
    from crdr_amd.hip import ops, lib as L
    name, n, h, w, ci, co = case
    dev = _dev()
    lib = L.load()
    ids = ops._stream_ids()
    assert len(ids) >= 1
    M = n * h * w
    ldx, ldy = ci + 32, co + 8
    X = _rand(M, ldx, seed=1).to(dev)
    wt = _rand(co, ci, 1, 1, seed=2, scale=ci ** -0.5)
    wp = ops.pack_weight(wt.to(dev), False)
    b = _rand(co, seed=3).to(dev)
    v2 = _rand(co, seed=4).to(dev)
    sc, sh = (_rand(co, seed=5) + 1.5).to(dev), _rand(co, seed=6).to(dev)
    RES = _rand(M, co + 4, seed=7).to(dev)
    PRE = _rand(M, co, seed=8).to(dev)
    MSK = _rand(M, co + 12, seed=9).to(dev)
    Y0 = _rand(M, ldy, seed=10).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    ref = (X[:, :ci].double().cpu() @ wt.view(co, ci).double().t())

    def run(algo, flags):
        d = L.ConvDesc(N=n, H=h, W=w, C=ci, OH=h, OW=w, OC=co, kh=1, kw=1, stride=1, pad=0, transposed=0, ldx=ldx, ldy=ldy,
                       wrows=wp.shape[1], wcols=wp.shape[2], flags=flags, ldres=co + 4, ldg=0, wlayout=0, reserved=algo,
                       ldpre=co, ldmask=co + 12)
        y = Y0.clone()
        io = L.ConvIO(x=X.data_ptr(), w=wp.data_ptr(), y=y.data_ptr(), bias=b.data_ptr(), vec2=v2.data_ptr(), res=RES.data_ptr(),
                      scale=sc.data_ptr(), shift=sh.data_ptr(), pre=PRE.data_ptr(), mask=MSK.data_ptr())
        cs = None
        if flags & L.EPI_COLSUM:
            rows, ld = C.c_int(), C.c_int()
            L.check(lib.crdr_conv2d_colsum_layout(C.byref(d), 1, C.byref(rows), C.byref(ld)), "layout")
            cs = torch.full((rows.value, 2, ld.value), float("nan"), device=dev)
            io.cs = cs.data_ptr()
        nb = lib.crdr_conv2d_workspace(C.byref(d))
        ws = torch.zeros(max(nb, 4) // 4, device=dev)   # split-K tickets at the head: zero on entry (CRDR_CONV_TICKETS)
        L.check(lib.crdr_conv2d(C.byref(d), C.byref(io), ws.data_ptr(), nb, s), f"conv2d algo {algo} flags {flags}")
        torch.cuda.synchronize()
        return y, cs

    FLAGSETS = [
        0,
        L.EPI_BIAS | L.EPI_RELU,
        L.EPI_BIAS | L.EPI_LRELU | L.EPI_VEC2,
        L.EPI_BIAS | L.EPI_RES | L.EPI_AFFINE,
        L.EPI_RELUMASK | L.EPI_COLSUM,
        L.EPI_RELUMASK | L.EPI_MASKOFF | L.EPI_COLSUM,
        L.EPI_BIAS | L.EPI_LRELUMASK | L.EPI_MASKOFF | L.EPI_COLSUM | L.EPI_RES,
        L.EPI_RES | L.EPI_COLSUM,
    ]
    ran = 0
    for algo in ids:
        try:
            y, _ = run(algo, 0)
        except L.CrdrHipError:
            continue  # this variant's weight tile does not fit LDS for this C (or too many column tiles)
        ran += 1
        _close(y[:, :co], ref, f"{name} stream {algo}")
        assert torch.equal(y[:, co:], Y0[:, co:]), "wrote outside its channel slice"
        for fl in FLAGSETS[1:]:
            y, cs = run(algo, fl)
            yt, cst = run(1, fl)  # tile config 0: 128 x 32
            assert torch.equal(y, yt), f"{name}: stream {algo} flags {fl} differs from the tiled kernel"
            if cs is not None:  # one partial row per 128- or 256-row tile
                assert torch.isfinite(cs[:, :, :co]).all()
                _close(cs[:, :, :co].sum(0), cst[:, :, :co].sum(0), f"{name} colsum {algo} flags {fl}", rtol=1e-5)
    assert ran >= 1, f"{name}: no streaming variant accepted"
    # and it refuses what it does not implement
    d = L.ConvDesc(N=1, H=8, W=8, C=64, OH=8, OW=8, OC=64, kh=3, kw=3, stride=1, pad=1, transposed=0, ldx=64, ldy=64, wrows=64,
                   wcols=64, flags=0, ldres=0, ldg=0, wlayout=0, reserved=ids[0], ldpre=0, ldmask=0)
    rows, ld = C.c_int(), C.c_int()
    assert lib.crdr_conv2d_colsum_layout(C.byref(d), 1, C.byref(rows), C.byref(ld)) != 0
    for bad in (L.EPI_PREADD, L.EPI_ACCUM, L.EPI_GATE):
        with pytest.raises(L.CrdrHipError):
            run(ids[0], bad)


def test_streaming_1x1_grouped():
    """A grouped launch (the two NLAM branches) through the persistent 1x1 kernel: every problem has its own tensors and
    column-sum rows; results equal the tiled kernel's bit for bit."""
    import ctypes as C
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    lib = L.load()
    n, h, w, ci, co, G = 2, 24, 20, 96, 192, 3
    M = n * h * w
    X = [_rand(M, ci, seed=10 + g).to(dev) for g in range(G)]
    W = [ops.pack_weight(_rand(co, ci, 1, 1, seed=20 + g, scale=ci ** -0.5).to(dev), False) for g in range(G)]
    B = [_rand(co, seed=30 + g).to(dev) for g in range(G)]
    R = [_rand(M, co, seed=40 + g).to(dev) for g in range(G)]
    K = [_rand(M, co, seed=50 + g).to(dev) for g in range(G)]
    s = torch.cuda.current_stream().cuda_stream

    def run(algo, flags):
        d = L.ConvDesc(N=n, H=h, W=w, C=ci, OH=h, OW=w, OC=co, kh=1, kw=1, stride=1, pad=0, transposed=0, ldx=ci, ldy=co,
                       wrows=W[0].shape[1], wcols=W[0].shape[2], flags=flags, ldres=co, ldg=0, wlayout=0, reserved=algo,
                       ldpre=0, ldmask=co)
        ys = [torch.zeros(M, co, device=dev) for _ in range(G)]
        rows, ld = C.c_int(), C.c_int()
        L.check(lib.crdr_conv2d_colsum_layout(C.byref(d), G, C.byref(rows), C.byref(ld)), "layout")
        css = [torch.full((max(rows.value, 1), 2, max(ld.value, 1)), float("nan"), device=dev) for _ in range(G)]
        ios = (L.ConvIO * G)()
        for g in range(G):
            ios[g].x, ios[g].w, ios[g].y = X[g].data_ptr(), W[g].data_ptr(), ys[g].data_ptr()
            ios[g].bias, ios[g].res, ios[g].mask, ios[g].cs = B[g].data_ptr(), R[g].data_ptr(), K[g].data_ptr(), css[g].data_ptr()
        nb = lib.crdr_conv2d_grouped_workspace(C.byref(d), G)
        ws = torch.zeros(max(nb, 4) // 4, device=dev)   # split-K tickets at the head: zero on entry (CRDR_CONV_TICKETS)
        L.check(lib.crdr_conv2d_grouped(C.byref(d), ios, G, ws.data_ptr(), nb, s), f"grouped algo {algo}")
        torch.cuda.synchronize()
        return ys, css

    ran = 0
    for algo in ops._stream_ids():
        for fl in (L.EPI_BIAS | L.EPI_RELU, L.EPI_BIAS | L.EPI_RES, L.EPI_RELUMASK | L.EPI_COLSUM | L.EPI_RES):
            try:
                ys, css = run(algo, fl)
            except L.CrdrHipError:
                continue
            ran += 1
            yt, cst = run(1, fl)
            for g in range(G):
                assert torch.equal(ys[g], yt[g]), f"problem {g} algo {algo} flags {fl}"
                if fl & L.EPI_COLSUM:
                    _close(css[g][:, :, :co].sum(0), cst[g][:, :, :co].sum(0), f"colsum problem {g}", rtol=1e-5)
            if fl == (L.EPI_BIAS | L.EPI_RES):
                ref = X[1].double().cpu() @ _rand(co, ci, 1, 1, seed=21, scale=ci ** -0.5).view(co, ci).double().t()
                _close(ys[1], ref + B[1].double().cpu() + R[1].double().cpu(), "grouped stream vs fp64")
    assert ran >= 3


def test_builtin_choice_streams_1x1_and_falls_back_when_unaligned():
    """Without a forced algorithm (the codec scripts run untuned) a large aligned 1x1 layer goes to the streaming kernel --
    same bits as the tiled kernel -- and an output pointer that is only 4-byte aligned takes the tiled kernel instead of
    failing (the built-in choice knows strides, not addresses)."""
    import ctypes as C
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    lib = L.load()
    n, h, w, ci, co = 4, 64, 64, 96, 192
    M = n * h * w
    X = _rand(M, ci, seed=1).to(dev)
    wt = _rand(co, ci, 1, 1, seed=2, scale=ci ** -0.5)
    wp = ops.pack_weight(wt.to(dev), False)
    b = _rand(co, seed=3).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    ref = X.double().cpu() @ wt.view(co, ci).double().t() + b.double().cpu()

    def run(algo, shift):
        d = L.ConvDesc(N=n, H=h, W=w, C=ci, OH=h, OW=w, OC=co, kh=1, kw=1, stride=1, pad=0, transposed=0, ldx=ci, ldy=co,
                       wrows=wp.shape[1], wcols=wp.shape[2], flags=L.EPI_BIAS, ldres=0, ldg=0, wlayout=0, reserved=algo,
                       ldpre=0, ldmask=0)
        buf = torch.zeros(M * co + 8, device=dev)
        io = L.ConvIO(x=X.data_ptr(), w=wp.data_ptr(), y=buf.data_ptr() + 4 * shift, bias=b.data_ptr())
        assert lib.crdr_conv2d_workspace(C.byref(d)) == 0  # the built-in choice for this layer needs no workspace
        L.check(lib.crdr_conv2d(C.byref(d), C.byref(io), None, 0, s), "conv2d")
        torch.cuda.synchronize()
        return buf[shift:shift + M * co].view(M, co).clone()

    y0 = run(0, 0)
    _close(y0, ref, "built-in choice, aligned")
    assert torch.equal(y0, run(1, 0)), "streaming and tiled kernels disagree"
    y1 = run(0, 1)
    _close(y1, ref, "built-in choice, output 4-byte aligned only")
    assert torch.equal(y1, y0)


def test_linear_group_matches_separate_layers():
    """crdr_linear_group_{fwd,bwd}: several projections of one conditioning vector in one launch per direction; outputs,
    the summed input gradient and the parameter gradients (accumulated into .grad) against fp64 torch."""
    from crdr_amd.hip import functional as HF
    from crdr_amd.models.layer.hip_layers import HipConv2d
    dev = _dev()
    torch.manual_seed(0)
    outs = [128, 128, 256, 96, 4, 33, 256, 128, 192]
    layers = [HipConv2d(512, o, 1).to(dev) for o in outs]
    x = _rand(1, 512, 1, 1, seed=1).to(dev).requires_grad_(True)
    cots = [_rand(1, o, seed=10 + k).to(dev) for k, o in enumerate(outs)]
    ys = HF.linear_group(x, layers)
    loss = sum((y * c).sum() for y, c in zip(ys, cots))
    loss.backward()
    xr = x.detach().double().cpu().reshape(1, 512).requires_grad_(True)
    tot = 0
    for k, ly in enumerate(layers):
        w = ly.weight.detach().double().cpu().reshape(outs[k], 512).requires_grad_(True)
        b = ly.bias.detach().double().cpu().requires_grad_(True)
        yr = xr @ w.t() + b
        _close(ys[k], yr, f"group fwd {k}")
        (yr * cots[k].double().cpu()).sum().backward()
        _close(ly.weight.grad.reshape(outs[k], 512), w.grad, f"group dW {k}")
        _close(ly.bias.grad, b.grad, f"group db {k}")
        tot += 1
    _close(x.grad.reshape(1, 512), xr.grad, "group dx (sum over the layers)")
    assert tot == len(outs)


def test_column_sums_every_tile_config():
    """CRDR_EPI_COLSUM (bias / beta-vector gradients out of the input-gradient conv) from every tile configuration, including
    those whose epilogue passes span three column blocks: sum over the partial rows == column sums of the stored output."""
    import ctypes as C
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    lib = L.load()
    n, h, w, ci, co, k = 2, 17, 19, 64, 224, 3
    M = n * h * w
    x = _rand(n, ci, h, w, seed=1).to(dev).contiguous(memory_format=torch.channels_last)
    xv, ldx = ops.nhwc(x)
    wt = _rand(co, ci, k, k, seed=2, scale=0.05)
    wp = ops.pack_weight(wt.to(dev), False)
    MSK = _rand(M, co, seed=3).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    for c in range(lib.crdr_conv2d_num_configs()):
        d = L.ConvDesc(N=n, H=h, W=w, C=ci, OH=h, OW=w, OC=co, kh=k, kw=k, stride=1, pad=1, transposed=0, ldx=ldx, ldy=co,
                       wrows=wp.shape[1], wcols=wp.shape[2], flags=L.EPI_RELUMASK | L.EPI_COLSUM, ldres=0, ldg=0, wlayout=0,
                       reserved=c + 1, ldpre=0, ldmask=co)
        rows, ld = C.c_int(), C.c_int()
        L.check(lib.crdr_conv2d_colsum_layout(C.byref(d), 1, C.byref(rows), C.byref(ld)), f"layout cfg {c}")
        y = torch.zeros(M, co, device=dev)
        cs = torch.full((rows.value, 2, ld.value), float("nan"), device=dev)
        io = L.ConvIO(x=xv.data_ptr(), w=wp.data_ptr(), y=y.data_ptr(), mask=MSK.data_ptr(), cs=cs.data_ptr())
        L.check(lib.crdr_conv2d(C.byref(d), C.byref(io), None, 0, s), f"conv2d cfg {c}")
        torch.cuda.synchronize()
        post = y.double().sum(0).cpu()
        pre = torch.nn.functional.conv2d(x.double().cpu(), wt.double(), padding=1).permute(0, 2, 3, 1).reshape(M, co).sum(0)
        _close(cs[:, 1, :co].sum(0), post, f"cfg {c}: column sums after the mask", rtol=1e-4)
        _close(cs[:, 0, :co].sum(0), pre, f"cfg {c}: column sums before the mask", rtol=1e-4)


@pytest.mark.parametrize("n,hh", [(16, 16), (2, 4)], ids=["M4096", "M32"])
def test_splitk_in_launch_reduce_is_exact_deterministic_and_self_cleaning(n, hh):
    """Split-K plans reduce inside the launch (igemm.hip: write-through slabs, ticket, last arriver adds the slabs in split order):
    every tile configuration x split depth on the Charm's 224->128 5x5 @16x16 shape, with every epilogue class (fast path,
    PREADD + bias + ReLU, ACCUM, mask + column sums), against fp64; 12 back-to-back launches on ONE workspace are bit-identical
    (the sum order is fixed, whoever arrives last) while a second stream keeps the chip unevenly loaded, and the tickets at the
    head of the workspace are zero again after every launch."""
    import ctypes as C
    from crdr_amd.hip import ops, lib as L
    dev = _dev()
    lib = L.load()
    ci, co, k = 224, 128, 5
    M = n * hh * hh
    x = _rand(n, ci, hh, hh, seed=1)
    wt = _rand(co, ci, k, k, seed=2, scale=(ci * k * k) ** -0.5)
    b = _rand(co, seed=3)
    ref = F.conv2d(x.double(), wt.double(), None, padding=2).permute(0, 2, 3, 1).reshape(M, co)
    X = x.permute(0, 2, 3, 1).reshape(M, ci).contiguous().to(dev)
    wp = ops.pack_weight(wt.to(dev), False)
    bd = b.to(dev)
    PRE = _rand(M, co, seed=8).to(dev)
    MSK = _rand(M, co, seed=9).to(dev)
    Y0 = _rand(M, co, seed=10).to(dev)
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    noise_a = torch.randn(4096, 4096, device=dev)
    cases = {
        "fast": (L.EPI_BIAS | L.EPI_RELU, lambda r: torch.relu(r + b.double())),
        "preadd": (L.EPI_PREADD | L.EPI_BIAS | L.EPI_RELU, lambda r: torch.relu(r + PRE.cpu().double() + b.double())),
        "accum": (L.EPI_ACCUM, lambda r: r + Y0.cpu().double()),
        "mask+colsum": (L.EPI_RELUMASK | L.EPI_COLSUM, lambda r: torch.where(MSK.cpu() > 0, r, torch.zeros_like(r))),
    }
    ran = 0
    for cfg in range(lib.crdr_conv2d_num_configs()):
        for ls in (1, 2, 3, 4):
            algo = (cfg + 1) | (ls << 8)
            for cname, (flags, fref) in cases.items():
                d = L.ConvDesc(N=n, H=hh, W=hh, C=ci, OH=hh, OW=hh, OC=co, kh=k, kw=k, stride=1, pad=2, transposed=0, ldx=ci, ldy=co,
                               wrows=wp.shape[1], wcols=wp.shape[2], flags=flags, ldres=0, ldg=0, wlayout=0, reserved=algo, ldpre=co, ldmask=co)
                nb = lib.crdr_conv2d_workspace(C.byref(d))
                if nb == 0:
                    continue   # the library rejects this combination (too few K tiles per split, too many tiles)
                if (ls, cname) not in ((3, "fast"), (3, "preadd"), (2, "accum"), (4, "mask+colsum"), (1, "mask+colsum")) and cfg % 5:
                    continue   # every configuration sees the main classes; the full cross product on every fifth
                ws = torch.zeros(nb // 4, device=dev)
                cs = None
                outs = []
                reps = (12 if cfg % 5 == 0 else 3) * (1 if M > 1000 else 3)
                for rep in range(reps):
                    y = Y0.clone()
                    io = L.ConvIO(x=X.data_ptr(), w=wp.data_ptr(), y=y.data_ptr(), bias=bd.data_ptr(), pre=PRE.data_ptr(), mask=MSK.data_ptr())
                    if flags & L.EPI_COLSUM:
                        rows, ld = C.c_int(), C.c_int()
                        L.check(lib.crdr_conv2d_colsum_layout(C.byref(d), 1, C.byref(rows), C.byref(ld)), "layout")
                        cs = torch.full((rows.value, 2, ld.value), float("nan"), device=dev)
                        io.cs = cs.data_ptr()
                    if rep % 2:   # uneven load: a GEMM on another stream competes for some of the CUs
                        with torch.cuda.stream(side):
                            torch.mm(noise_a[: 512 * (1 + rep % 3)], noise_a)
                    L.check(lib.crdr_conv2d(C.byref(d), C.byref(io), ws.data_ptr(), nb, main.cuda_stream), f"conv2d cfg {cfg} split {1 << ls} {cname}")
                    outs.append((y, cs))
                torch.cuda.synchronize()
                assert int(ws[:16384].view(torch.int32).abs().max()) == 0, f"tickets not back at zero (cfg {cfg} split {1 << ls} {cname})"
                _close(outs[0][0], fref(ref), f"cfg {cfg} split {1 << ls} {cname}")
                for y, c_ in outs[1:]:
                    assert torch.equal(y, outs[0][0]), f"cfg {cfg} split {1 << ls} {cname}: launches differ"
                    if c_ is not None:
                        assert torch.equal(c_, outs[0][1])
                if cs is not None:
                    got = outs[0][1].double().cpu()
                    want_pre, want_post = ref.sum(0), fref(ref).sum(0)
                    scale = ref.abs().sum(0).max().item()
                    assert (got[:, 0, :co].sum(0) - want_pre).abs().max().item() <= 2e-5 * scale
                    assert (got[:, 1, :co].sum(0) - want_post).abs().max().item() <= 2e-5 * scale
                ran += 1
    assert ran >= 3 * lib.crdr_conv2d_num_configs(), ran
