"""Winograd F(2x2, 3x3) kernel (csrc/wino.hip, forced algorithm id) against fp64 torch and against the implicit-GEMM kernel:
forward and input gradient of 3x3 stride-1 convolutions, ragged sizes (partial 16x16 patches, odd H / W), channel counts that
are not multiples of 8 / 32 / 64, channel-slice strides, and the fused epilogues.  fp32 arithmetic with a different association
of the sums: the deviation from fp64 is held to the same bound as the direct kernel's (rtol 2e-4 of the output scale) and the
measured ratio of the two kernels' errors is printed."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_conv import _close, _dev, _rand

pytestmark = pytest.mark.gpu


def _wino_id():
    from crdr_amd.hip import lib as L
    lib = L.load()
    return lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs()


CASES = [
    # name, N, Cin, H, W, Cout, pad[, k]
    ("96_96_16", 2, 96, 16, 16, 96, 1),
    ("128_128_ragged", 1, 128, 34, 38, 128, 1),
    ("64_160_odd", 2, 64, 15, 15, 160, 1),
    ("36_40_c4", 1, 36, 20, 20, 40, 1),
    ("256_512_32", 1, 256, 32, 32, 512, 1),
    ("32_64_valid", 1, 32, 18, 21, 64, 0),
    ("8_8_tiny", 1, 8, 5, 3, 8, 1),
    ("64_96_9patches", 1, 64, 34, 38, 96, 1),     # channel tail of 32: pair tiles, odd patch count (the last pair has one patch)
    ("32_224_pairs", 3, 32, 16, 16, 224, 1),
    # 5x5 = 2 x 2 sub-filters of 3x3 (Charm transforms, minnen20_charm_context_model.py:29-35)
    ("k5_320_224_16", 2, 320, 16, 16, 224, 2, 5),
    ("k5_32_96_ragged", 1, 32, 19, 23, 96, 2, 5),
    ("k5_36_40_valid", 1, 36, 12, 9, 40, 0, 5),
    ("k5_64_64_pad1", 1, 64, 10, 10, 64, 1, 5),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_winograd_fwd_dgrad(case):
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, p = case[:7]
    k = case[7] if len(case) > 7 else 3
    dev = _dev()
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, k, k, seed=2, scale=(ci * k * k) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, wt.double(), b.double(), padding=p)
    oh, ow = ref.shape[2:]
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=False)
    direct = ops.conv2d_raw(xd, wp, co, (k, k), 1, p, False, (oh, ow), bias=bd, flags=1, algo=1)
    out = ops.conv2d_raw(xd, wp, co, (k, k), 1, p, False, (oh, ow), bias=bd, flags=1, algo=_wino_id())
    torch.cuda.synchronize()
    _close(out, ref, name + " fwd")
    e_w = (out.cpu().double() - ref.detach()).abs().max().item()
    e_d = (direct.cpu().double() - ref.detach()).abs().max().item()
    print(f"{name}: fwd max err winograd {e_w:.2e} direct {e_d:.2e} (scale {ref.abs().max().item():.2e})")
    wq = ops.pack_weight(wd, transpose=True)
    dx = ops.conv2d_raw(dyd, wq, ci, (k, k), 1, p, True, (h, w), algo=_wino_id())
    _close(dx, xr.grad, name + " dgrad")
    # deterministic
    out2 = ops.conv2d_raw(xd, wp, co, (k, k), 1, p, False, (oh, ow), bias=bd, flags=1, algo=_wino_id())
    assert torch.equal(out, out2)
    # variant 1 (pair tiles for a channel tail of <= 32): same products in the same order, bit for bit the same result
    from crdr_amd.hip import lib as L
    patches = lambda a, b: n * ((a + 15) // 16) * ((b + 15) // 16)
    for what, cc, hw, run, want in (("fwd", co, (oh, ow), lambda a: ops.conv2d_raw(xd, wp, co, (k, k), 1, p, False, (oh, ow), bias=bd, flags=1, algo=a), out),
                                    ("dgrad", ci, (h, w), lambda a: ops.conv2d_raw(dyd, wq, ci, (k, k), 1, p, True, (h, w), algo=a), dx)):
        if 0 < cc % 64 <= 32 and patches(*hw) > 1:
            assert torch.equal(run(_wino_id() + 1), want), f"{name} {what}: pair-tile variant differs"
        else:
            with pytest.raises(L.CrdrHipError):
                run(_wino_id() + 1)


def test_winograd_epilogues_and_slices():
    """bias + ReLU + vec2 + residual + affine, output written into a channel slice of a wider tensor, input read from one."""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    dev = _dev()
    n, ci, h, w, co = 2, 64, 24, 20, 96
    wide = _rand(n, ci + 32, h, w, seed=5).to(dev).contiguous(memory_format=torch.channels_last)
    x = wide[:, 16:16 + ci]
    wt = _rand(co, ci, 3, 3, seed=6, scale=(ci * 9) ** -0.5).to(dev)
    b, v2, sc, sh = (_rand(co, seed=s).to(dev) for s in (7, 8, 9, 10))
    res = _rand(n, co, h, w, seed=11).to(dev).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_weight(wt, transpose=False)
    flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_VEC2 | L.EPI_RES | L.EPI_AFFINE
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1).relu() + v2.double().view(1, -1, 1, 1) + res.double()
    ref = ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    owide = torch.zeros(n, co + 40, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.conv2d_raw(x, wp, co, (3, 3), 1, 1, False, (h, w), bias=b, flags=flags, vec2=v2, res=res, scale=sc, shift=sh,
                         out=owide[:, 8:8 + co], algo=_wino_id())
    torch.cuda.synchronize()
    _close(out, ref, "epilogues")
    assert float(owide[:, :8].abs().max()) == 0.0 and float(owide[:, 8 + co:].abs().max()) == 0.0
    # LeakyReLU
    out = ops.conv2d_raw(x, wp, co, (3, 3), 1, 1, False, (h, w), bias=b, flags=L.EPI_BIAS | L.EPI_LRELU, algo=_wino_id())
    _close(out, F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2), "lrelu")


W4_CASES = [
    # name, N, Cin, H, W, Cout, pad   (F(4x4, 3x3): output tiles of 8 x 64 pixels x 64 channels beyond 32 output columns, of 16 x 32 pixels at 24..32)
    ("w4_96_96_64", 2, 96, 64, 64, 96, 1),
    ("w4_128_128_ragged", 1, 128, 21, 70, 128, 1),      # partial tiles on both edges, 21 = 2 * 8 + 5 rows, 70 = 64 + 6 columns
    ("w4_36_40_c4", 1, 36, 9, 50, 40, 1),               # channel tail of 4 (half a chunk), output channels not a multiple of 32
    ("w4_64_160_valid", 1, 64, 18, 66, 160, 0),         # pad 0: 16 x 64 outputs
    ("w4_256_64_128", 1, 256, 16, 128, 64, 1),          # two tile columns
    ("w4_8_8_wide", 3, 8, 5, 48, 8, 1),
    ("w4_128_128_32", 2, 128, 32, 32, 128, 1),          # the 16 x 32 tile geometry: two tiles per image
    ("w4_96_72_narrow_ragged", 3, 96, 21, 27, 72, 1),   # partial tiles both ways (21 = 16 + 5 rows, 27 of 32 columns)
    ("w4_64_64_valid_30", 1, 64, 34, 32, 64, 0),        # pad 0: 32 x 30 outputs
    ("w4_40_36_pad2_28", 1, 40, 12, 28, 36, 2),         # pad 2 (the transposed twin of pad 0): 14 x 30 outputs
    ("w4_160_160_16", 5, 160, 16, 16, 160, 1),          # 16 x 16 images, two per tile (odd batch: the last tile holds one)
    ("w4_64_72_13x11", 3, 64, 13, 11, 72, 1),           # ragged images inside the 16 x 16 frame
    ("w4_32_32_valid16", 2, 32, 16, 16, 32, 0),         # pad 0: 14 x 14 outputs
]


@pytest.mark.parametrize("case", W4_CASES, ids=[c[0] for c in W4_CASES])
def test_winograd_f4x4_fwd_dgrad(case):
    """Winograd F(4x4, 3x3) kernel (csrc/wino4.hip, Winograd variant 2) against fp64 torch: forward and input gradient.  Its transforms
    carry the constants 4, 5, 8, 1/24: the deviation from fp64 is ~8x the direct kernel's (numpy model: 5e-6 .. 1e-5 of the output
    scale at 96 .. 256 channels) and is gated at 2e-5 (round 5: interpolation points 0, +-3/4, +-5/4; the 5e-5 of round 4 belonged to 0, +-1, +-2); the measured errors of the three kernels are printed."""
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, p = case
    dev = _dev()
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 3, 3, seed=2, scale=(ci * 9) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, wt.double(), b.double(), padding=p)
    oh, ow = ref.shape[2:]
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=False)
    a4 = _wino_id() + 2
    direct = ops.conv2d_raw(xd, wp, co, (3, 3), 1, p, False, (oh, ow), bias=bd, flags=1, algo=1)
    w2 = ops.conv2d_raw(xd, wp, co, (3, 3), 1, p, False, (oh, ow), bias=bd, flags=1, algo=_wino_id())
    out = ops.conv2d_raw(xd, wp, co, (3, 3), 1, p, False, (oh, ow), bias=bd, flags=1, algo=a4)
    torch.cuda.synchronize()
    sc = ref.abs().max().item()
    e4 = (out.cpu().double() - ref.detach()).abs().max().item() / sc
    e2 = (w2.cpu().double() - ref.detach()).abs().max().item() / sc
    ed = (direct.cpu().double() - ref.detach()).abs().max().item() / sc
    print(f"{name}: fwd max err / scale: F(4x4) {e4:.2e}  F(2x2) {e2:.2e}  direct {ed:.2e}")
    _close(out, ref, name + " fwd", rtol=2e-5)
    wq = ops.pack_weight(wd, transpose=True)
    dx = ops.conv2d_raw(dyd, wq, ci, (3, 3), 1, p, True, (h, w), algo=a4) if (w >= 24 or (9 <= w <= 16 and 9 <= h <= 16)) else None
    if dx is not None:
        _close(dx, xr.grad, name + " dgrad", rtol=2e-5)
    out2 = ops.conv2d_raw(xd, wp, co, (3, 3), 1, p, False, (oh, ow), bias=bd, flags=1, algo=a4)
    assert torch.equal(out, out2), "not deterministic"


W4S2_CASES = [
    # name, N, Cin, H, W, Cout   (5x5 stride 2 pad 2; conv: input H x W -> H/2 x W/2; transposed: input H x W -> 2H x 2W)
    ("w4s2_192_192_128", 1, 192, 32, 128, 192),
    ("w4s2_64_96_ragged", 2, 64, 20, 104, 96),        # 10 x 52 outputs: partial tiles both ways
    ("w4s2_36_40_c4", 1, 36, 16, 96, 40),
    ("w4s2_256_256_wide", 1, 256, 16, 256, 256),
    ("w4s2_192_192_64", 2, 192, 64, 64, 192),         # 32 x 32 outputs / phases of 64 x 64: the 16 x 32 tile geometry for the conv
    ("w4s2_96_64_narrow", 1, 96, 36, 28, 64),         # transposed: phases of 36 x 28 (ragged 16 x 32 tiles); conv: 14 output columns, not taken
    ("w4s2_64_128_c56", 3, 64, 24, 56, 128),          # conv: 12 x 28 outputs
]


@pytest.mark.parametrize("case", W4S2_CASES, ids=[c[0] for c in W4S2_CASES])
def test_winograd_f4x4_5x5_stride2(case):
    """the 5x5 stride-2 layers through the F(4x4, 3x3) kernel: Conv2d 5x5 s2 p2 as four parity sub-filters accumulated over the parity
    planes of the input (elic_autoencoder.py:42-52), ConvTranspose2d 5x5 s2 p2 op1 as four output phases (elic_layers.py:14-21), and the
    input gradient of each (= the other form), against fp64 torch"""
    from crdr_amd.hip import ops
    name, n, ci, h, w, co = case
    dev = _dev()
    a4 = _wino_id() + 2
    # Conv2d 5x5 s2: forward + input gradient
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 5, 5, seed=2, scale=(ci * 25 / 4) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, wt.double(), b.double(), stride=2, padding=2)
    oh, ow = ref.shape[2:]
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=False)
    if ow >= 24:
        direct = ops.conv2d_raw(xd, wp, co, (5, 5), 2, 2, False, (oh, ow), bias=bd, flags=1, algo=1)
        out = ops.conv2d_raw(xd, wp, co, (5, 5), 2, 2, False, (oh, ow), bias=bd, flags=1, algo=a4)
        torch.cuda.synchronize()
        sc = ref.abs().max().item()
        print(f"{name}: conv s2 fwd max err / scale: F(4x4) {(out.cpu().double() - ref.detach()).abs().max().item() / sc:.2e}  "
              f"direct {(direct.cpu().double() - ref.detach()).abs().max().item() / sc:.2e}")
        _close(out, ref, name + " conv fwd", rtol=2e-5)
        assert torch.equal(out, ops.conv2d_raw(xd, wp, co, (5, 5), 2, 2, False, (oh, ow), bias=bd, flags=1, algo=a4))
        # its input gradient: a transposed stride-2 launch (output = the conv's input grid, phases of oh x ow)
        wq = ops.pack_weight(wd, transpose=True)
        dx = ops.conv2d_raw(dyd, wq, ci, (5, 5), 2, 2, True, (h, w), algo=a4)
        _close(dx, xr.grad, name + " conv dgrad", rtol=2e-5)
    # ConvTranspose2d 5x5 s2 p2 op1: forward + input gradient
    if w >= 24:
        wt2 = _rand(ci, co, 5, 5, seed=5, scale=(ci * 25 / 4) ** -0.5)
        xr2 = x.double().requires_grad_(True)
        ref2 = F.conv_transpose2d(xr2, wt2.double(), b.double(), stride=2, padding=2, output_padding=1)
        oh2, ow2 = ref2.shape[2:]
        dy2 = _rand(*ref2.shape, seed=6)
        ref2.backward(dy2.double())
        w2d = wt2.to(dev)
        wp2 = ops.pack_weight(w2d, transpose=True)     # rows = Cout, cols = Cin
        out2 = ops.conv2d_raw(xd, wp2, co, (5, 5), 2, 2, True, (oh2, ow2), bias=bd, flags=1, algo=a4)
        _close(out2, ref2, name + " convT fwd", rtol=2e-5)
        wq2 = ops.pack_weight(w2d, transpose=False)    # rows = Cin, cols = Cout
        dx2 = ops.conv2d_raw(dy2.to(dev), wq2, ci, (5, 5), 2, 2, False, (h, w), algo=a4)
        _close(dx2, xr2.grad, name + " convT dgrad", rtol=2e-5)


W4K5_CASES = [
    # name, N, Cin, H, W, Cout   (5x5 stride 1 pad 2 as four shifted 3x3 sub-filters)
    ("w4k5_320_128_16", 5, 320, 16, 16, 128),         # the context model's stage: two 16 x 16 images per tile, odd batch
    ("w4k5_32_96_16", 3, 32, 16, 16, 96),
    ("w4k5_64_64_12x14", 2, 64, 12, 14, 64),          # ragged inside the 16 x 16 frame
    ("w4k5_96_64_20x70", 1, 96, 20, 70, 64),          # 8 x 64 tiles, ragged both ways
    ("w4k5_48_80_24x32", 2, 48, 24, 32, 80),          # 16 x 32 tiles
    ("w4k5_12_8_min", 1, 12, 9, 9, 8),                # smallest channel count taken
]


@pytest.mark.parametrize("case", W4K5_CASES, ids=[c[0] for c in W4K5_CASES])
def test_winograd_f4x4_5x5_stride1(case):
    """the 5x5 stride-1 layers (the slice transforms of the context model, minnen20_charm_context_model.py:26-38) through the F(4x4, 3x3)
    kernel: taps (3 bi + a, 3 bj + b), four sub-filters accumulated over patches displaced by (3 bi, 3 bj); forward and input gradient (the
    transposed twin) against fp64 torch, bit-identical run to run"""
    from crdr_amd.hip import ops
    name, n, ci, h, w, co = case
    dev = _dev()
    a4 = _wino_id() + 2
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 5, 5, seed=2, scale=(ci * 25) ** -0.5)
    b = _rand(co, seed=3)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, wt.double(), b.double(), padding=2)
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), wt.to(dev), b.to(dev), dy.to(dev)
    wp = ops.pack_weight(wd, transpose=False)
    direct = ops.conv2d_raw(xd, wp, co, (5, 5), 1, 2, False, (h, w), bias=bd, flags=1, algo=1)
    out = ops.conv2d_raw(xd, wp, co, (5, 5), 1, 2, False, (h, w), bias=bd, flags=1, algo=a4)
    torch.cuda.synchronize()
    sc = ref.abs().max().item()
    print(f"{name}: fwd max err / scale: F(4x4) {(out.cpu().double() - ref.detach()).abs().max().item() / sc:.2e}  "
          f"direct {(direct.cpu().double() - ref.detach()).abs().max().item() / sc:.2e}")
    _close(out, ref, name + " fwd", rtol=2e-5)
    assert torch.equal(out, ops.conv2d_raw(xd, wp, co, (5, 5), 1, 2, False, (h, w), bias=bd, flags=1, algo=a4))
    if co >= 12:
        wq = ops.pack_weight(wd, transpose=True)
        dx = ops.conv2d_raw(dyd, wq, ci, (5, 5), 1, 2, True, (h, w), algo=a4)
        _close(dx, xr.grad, name + " dgrad", rtol=2e-5)


WG5_CASES = [
    # name, N, Cin, H, W, Cout, stride, log2 split   (5x5 pad 2 weight gradients through the F(3x3, 4x4) slab kernel: four 3x3 sub-problems)
    ("wg5_s1_hoist", 4, 64, 16, 16, 160, 1, 0),       # the context model's stage
    ("wg5_s1_ragged", 2, 36, 13, 22, 40, 1, 1),
    ("wg5_s1_wide", 1, 32, 9, 70, 64, 1, 2),
    ("wg5_s2_192", 2, 96, 32, 32, 96, 2, 1),           # Conv2d 5x5 s2: P = dy at 16 x 16, Q = x at 32 x 32
    ("wg5_s2_ragged", 1, 40, 20, 44, 72, 2, 0),
    ("wg5_s2_c4", 3, 12, 16, 16, 8, 2, 0),
]


@pytest.mark.parametrize("case", WG5_CASES, ids=[c[0] for c in WG5_CASES])
def test_winograd_wgrad_5x5(case):
    """5x5 weight gradients through wino4_wgrad.hip: stride 1 as four shifted sub-filters (taps 3 bi + a, 3 bj + b), stride 2 as four parity
    sub-filters (taps 2 a + ph, 2 b + pw) over the parity planes of the larger operand -- Conv2d (P = dy) and ConvTranspose2d (P = x) -- vs
    fp64 torch, with accumulation, bit-identical run to run"""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, st, ls = case
    dev = _dev()
    algo = (L.load().crdr_conv2d_wgrad_num_configs() + 1) | (ls << 8)
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 5, 5, seed=2, scale=(ci * 25) ** -0.5)
    xr, wr = x.double(), wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=st, padding=2)
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    g = torch.zeros(co, ci, 5, 5, device=dev)
    ops.conv2d_wgrad_raw(dyd, xd, g, (5, 5), st, 2, accumulate=False, algo=algo)
    gd = torch.zeros_like(g)
    ops.conv2d_wgrad_raw(dyd, xd, gd, (5, 5), st, 2, accumulate=False, algo=1)
    sc = wr.grad.abs().max().item()
    print(f"{name}: wgrad max err / scale: F(3x3,4x4) {(g.cpu().double() - wr.grad).abs().max().item() / sc:.2e}  direct {(gd.cpu().double() - wr.grad).abs().max().item() / sc:.2e}")
    _close(g, wr.grad, name + " wgrad", rtol=2e-5)
    g2 = torch.zeros_like(g)
    ops.conv2d_wgrad_raw(dyd, xd, g2, (5, 5), st, 2, accumulate=False, algo=algo)
    assert torch.equal(g, g2)
    ops.conv2d_wgrad_raw(dyd, xd, g, (5, 5), st, 2, accumulate=True, algo=algo)
    _close(g, 2 * wr.grad, name + " wgrad accumulate", rtol=2e-5)
    if st == 2 and h % 2 == 0 and w % 2 == 0:   # ConvTranspose2d 5x5 s2 p2 op1: P = x, Q = dy (twice the size)
        wT = _rand(ci, co, 5, 5, seed=5, scale=(ci * 25 / 4) ** -0.5).double().requires_grad_(True)
        rT = F.conv_transpose2d(xr, wT, None, stride=2, padding=2, output_padding=1)
        dyT = _rand(*rT.shape, seed=6)
        rT.backward(dyT.double())
        gT = torch.zeros(ci, co, 5, 5, device=dev)
        ops.conv2d_wgrad_raw(xd, dyT.to(dev), gT, (5, 5), 2, 2, accumulate=False, algo=algo)
        _close(gT, wT.grad, name + " convT wgrad", rtol=2e-5)


W4SPLIT_CASES = [
    # name, N, Cin, H, W, Cout, k, transposed, splits   (K splits inside the launch: partial tiles through slabs, last arriver reduces)
    ("w4sp_hoist_dgrad", 6, 608, 16, 16, 64, 5, 1, 6),      # the shape class it is for: many input channels, few tiles (5x5, two images per tile)
    ("w4sp_k5_c24_s2", 3, 24, 16, 16, 96, 5, 0, 2),          # splits inside and across the shifted sub-filters
    ("w4sp_k5_c48_s8", 2, 48, 12, 14, 40, 5, 0, 8),
    ("w4sp_k3_256_32", 2, 256, 32, 32, 128, 3, 0, 3),        # 3x3, 16 x 32 tiles, uneven split (64 chunks / 3)
    ("w4sp_k3_96_wide", 1, 96, 9, 70, 72, 3, 0, 4),          # 8 x 64 tiles, ragged
    ("w4sp_k5s2_conv", 1, 64, 16, 96, 64, -5, 0, 4),         # 5x5 stride-2 conv (parity sub-filters), k = -5 marks stride 2
    ("w4sp_k5s2_convT", 1, 64, 8, 48, 64, -5, 1, 2),         # 5x5 stride-2 transposed conv (phases)
]


@pytest.mark.parametrize("case", W4SPLIT_CASES, ids=[c[0] for c in W4SPLIT_CASES])
def test_winograd_f4x4_split_k(case):
    """the F(4x4) kernel with K splits (forced id | (splits - 1) << 8): equal to float64 within the kernel's bound, to the unsplit launch
    within fp32 summation order, bit-identical run to run (fixed split order), tickets left at zero (a second launch works), with a
    bias + ReLU epilogue and column sums taken by the last arriver"""
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, k, tr, ns = case
    dev = _dev()
    stride = 2 if k < 0 else 1
    k = abs(k)
    a4 = _wino_id() + 2
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(*((ci, co, k, k) if tr else (co, ci, k, k)), seed=2, scale=(ci * k * k / stride ** 2) ** -0.5)
    b = _rand(co, seed=3)
    if tr:
        ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=stride, padding=k // 2, output_padding=stride - 1)
    else:
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=k // 2)
    ref = torch.relu(ref)
    oh, ow = ref.shape[2:]
    xd, bd = x.to(dev), b.to(dev)
    wp = ops.pack_weight(wt.to(dev), transpose=bool(tr))
    one = ops.conv2d_raw(xd, wp, co, (k, k), stride, k // 2, bool(tr), (oh, ow), bias=bd, flags=3, algo=a4)
    outs = [ops.conv2d_raw(xd, wp, co, (k, k), stride, k // 2, bool(tr), (oh, ow), bias=bd, flags=3, algo=a4 | ((ns - 1) << 8)) for _ in range(3)]
    torch.cuda.synchronize()
    _close(outs[0], ref, name + " split vs fp64", rtol=2e-5)
    _close(outs[0], one.double(), name + " split vs unsplit", rtol=2e-5)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split launches differ run to run"
    after = ops.conv2d_raw(xd, wp, co, (k, k), stride, k // 2, bool(tr), (oh, ow), bias=bd, flags=3, algo=a4)
    assert torch.equal(after, one), "an unsplit launch after split ones differs (tickets / workspace)"


def test_winograd_f4x4_persistent_filter_caches():
    """ops' persistent filter caches: a second F(4x4) launch with the same registered weight pack reuses the transformed filters of the first
    (same bits out), inside or outside a filter_scope; a pack rewritten in place is noticed through its version counter alone; temporary
    packs are never cached; and the batched rebuild behind a pack refill (ops.FilterTable: one launch for every cache derived from a set of
    packs) leaves exactly what the per-launch transform would have written, stamped current"""
    from crdr_amd.hip import ops
    dev = _dev()
    a4 = _wino_id() + 2
    x = _rand(2, 96, 16, 64, seed=1).to(dev)
    x5 = _rand(2, 32, 16, 16, seed=4).to(dev)
    w1, w2 = _rand(64, 96, 3, 3, seed=2, scale=0.03).to(dev), _rand(64, 96, 3, 3, seed=3, scale=0.03).to(dev)
    w5a, w5b = _rand(96, 32, 5, 5, seed=5, scale=0.03).to(dev), _rand(96, 32, 5, 5, seed=6, scale=0.03).to(dev)
    wp = ops.pack_weight(w1, transpose=False)
    wp5 = ops.pack_weight(w5a, transpose=False)
    call = lambda: ops.conv2d_raw(x, wp, 64, (3, 3), 1, 1, False, (16, 64), algo=a4)
    call5 = lambda: ops.conv2d_raw(x5, wp5, 96, (5, 5), 1, 2, False, (16, 16), algo=a4)
    ref1, ref5a = call(), call5()
    st = ops.FILTER_SCOPE_STATS
    f0, r0 = st["filled"], st["reused"]
    assert torch.equal(call(), ref1) and (st["filled"], st["reused"]) == (f0, r0), "a temporary pack must not be cached by address"
    ops.register_persistent_pack(wp)
    ops.register_persistent_pack(wp5)
    assert torch.equal(call(), ref1) and (st["filled"], st["reused"]) == (f0 + 1, r0)
    with ops.filter_scope():
        assert torch.equal(call(), ref1) and (st["filled"], st["reused"]) == (f0 + 1, r0 + 1)
    assert torch.equal(call(), ref1) and (st["filled"], st["reused"]) == (f0 + 1, r0 + 2)
    # a writer that only bumps the pack's version (what functional._PackEntry.fill / PackTable.refill do): the kept filters are NOT reused
    wp.copy_(ops.pack_weight(w2, transpose=False))
    ops.bump_pack_version(wp.data_ptr())
    out2 = call()
    assert (st["filled"], st["reused"]) == (f0 + 2, r0 + 2)
    assert torch.equal(call(), out2) and (st["filled"], st["reused"]) == (f0 + 2, r0 + 3)
    _close(out2, F.conv2d(x.cpu().double(), w2.cpu().double(), padding=1), "filter cache after a refill", rtol=2e-5)
    # the batched rebuild: both packs rewritten, ONE launch rebuilds both caches (3x3: one sub-filter; 5x5 stride 1: four), the next
    # launches trust them and give the bits a fresh per-launch transform gives
    assert torch.equal(call5(), ref5a)                                    # (creates the 5x5 cache)
    wp.copy_(ops.pack_weight(w1, transpose=False))
    wp5.copy_(ops.pack_weight(w5b, transpose=False))
    ops.bump_pack_version(wp.data_ptr())
    ops.bump_pack_version(wp5.data_ptr())
    table = ops.FilterTable(dev)
    b0 = st["batched"]
    table.refill({wp.data_ptr(), wp5.data_ptr()})
    assert st["batched"] == b0 + 2 and len(table.entries) == 2
    f1, r1 = st["filled"], st["reused"]
    got3, got5 = call(), call5()
    assert (st["filled"], st["reused"]) == (f1, r1 + 2), "the launches behind the batched rebuild transformed again"
    assert torch.equal(got3, ref1)
    fresh5 = ops.conv2d_raw(x5, ops.pack_weight(w5b, transpose=False), 96, (5, 5), 1, 2, False, (16, 16), algo=a4)   # temporary pack: per-launch transform
    assert torch.equal(got5, fresh5)
    _close(got5, F.conv2d(x5.cpu().double(), w5b.cpu().double(), padding=2), "5x5 behind the batched rebuild", rtol=2e-5)
    # a rebuild for OTHER packs leaves these caches alone; rebuilding again is idempotent
    table.refill({wp.data_ptr(), wp5.data_ptr()})
    assert torch.equal(call(), ref1) and torch.equal(call5(), fresh5)
    ops.filter_scope_invalidate(wp.data_ptr())
    ops.filter_scope_invalidate(wp5.data_ptr())


def test_winograd_f4x4_epilogues_slices_groups_colsum():
    """every epilogue the F(4x4) kernel takes: bias + ReLU + vec2 + residual + affine, LeakyReLU, accumulate, ReLU-mask with column sums (the
    input-gradient launches of a conv chain), channel slices in and out, a grouped launch"""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    dev = _dev()
    a4 = _wino_id() + 2
    n, ci, h, w, co = 2, 64, 20, 72, 96
    wide = _rand(n, ci + 32, h, w, seed=5).to(dev).contiguous(memory_format=torch.channels_last)
    x = wide[:, 16:16 + ci]
    wt = _rand(co, ci, 3, 3, seed=6, scale=(ci * 9) ** -0.5).to(dev)
    b, v2, sc, sh = (_rand(co, seed=s).to(dev) for s in (7, 8, 9, 10))
    res = _rand(n, co, h, w, seed=11).to(dev).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_weight(wt, transpose=False)
    z = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    view = lambda t: t.double().view(1, -1, 1, 1)
    flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_VEC2 | L.EPI_RES | L.EPI_AFFINE
    ref = (z.relu() + view(v2) + res.double()) * view(sc) + view(sh)
    owide = torch.zeros(n, co + 40, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    out = ops.conv2d_raw(x, wp, co, (3, 3), 1, 1, False, (h, w), bias=b, flags=flags, vec2=v2, res=res, scale=sc, shift=sh,
                         out=owide[:, 8:8 + co], algo=a4)
    torch.cuda.synchronize()
    _close(out, ref, "epilogues", rtol=2e-5)
    assert float(owide[:, :8].abs().max()) == 0.0 and float(owide[:, 8 + co:].abs().max()) == 0.0
    out = ops.conv2d_raw(x, wp, co, (3, 3), 1, 1, False, (h, w), bias=b, flags=L.EPI_BIAS | L.EPI_LRELU, algo=a4)
    _close(out, F.leaky_relu(z, 0.2), "lrelu", rtol=2e-5)
    # accumulate into an existing tensor
    acc0 = _rand(n, co, h, w, seed=12).to(dev).contiguous(memory_format=torch.channels_last)
    acc = acc0.clone()
    ops.conv2d_raw(x, wp, co, (3, 3), 1, 1, False, (h, w), bias=b, flags=L.EPI_BIAS | L.EPI_ACCUM, out=acc, algo=a4)
    _close(acc, z + acc0.double(), "accumulate", rtol=2e-5)
    # grouped launch with ReLU mask + column sums (conv_multi: what a chain's input-gradient launches look like), against the built-in plan
    G = 2
    xs = [_rand(n, h, w, ci, seed=20 + g).to(dev) for g in range(G)]
    ws_ = [ops.pack_weight(_rand(co, ci, 3, 3, seed=30 + g, scale=(ci * 9) ** -0.5).to(dev), transpose=False) for g in range(G)]
    masks = [_rand(n, h, w, co, seed=40 + g).to(dev) for g in range(G)]

    def run(algo):
        keep = ops.FORCED_CONV_ALGO
        ops.FORCED_CONV_ALGO = algo
        try:
            ys = [torch.zeros(n, h, w, co, device=dev) for _ in range(G)]
            q = ops.colsum_queue(dev)
            r = ops.conv_multi(n, h, w, h, w, [ops.view(t, 0, ci) for t in xs], [t.data_ptr() for t in ws_], [ops.view(t, 0, co) for t in ys], co,
                               (3, 3), 1, 1, False, wrows=ws_[0].shape[1], wcols=ws_[0].shape[2], masks=[ops.view(t, 0, co) for t in masks],
                               flags=L.EPI_RELUMASK, colsum=True, device=dev)
            cs = []
            for ptr, rows, ld in r:
                off = ptr - q.arena.data_ptr()
                cs.append(q.arena[off:off + rows * 2 * ld * 4].view(torch.float32).view(rows, 2, ld).double().sum(0)[:, :co].clone())
            q.off, q.jobs, q.scratch_off = 0, [], 0
            torch.cuda.synchronize()
            return ys, cs
        finally:
            ops.FORCED_CONV_ALGO = keep
    yb, cb = run(1)
    y4, c4 = run(a4)
    for g in range(G):
        _close(y4[g], yb[g], f"grouped masked output {g}", rtol=2e-5)
        _close(c4[g], cb[g], f"grouped column sums {g}", rtol=2e-5)
        refg = F.conv2d(xs[g].permute(0, 3, 1, 2).double().cpu(), ws_[g][:, :co, :ci].permute(1, 2, 0).reshape(co, ci, 3, 3).double().cpu(), padding=1)
        refg = torch.where(masks[g].permute(0, 3, 1, 2).cpu() > 0, refg, torch.zeros_like(refg))
        _close(y4[g].permute(0, 3, 1, 2), refg, f"grouped masked output {g} vs fp64", rtol=2e-5)


def test_winograd_rejects_other_shapes():
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    dev = _dev()
    x = _rand(1, 32, 8, 8, seed=1).to(dev)
    w7 = ops.pack_weight(_rand(32, 32, 7, 7, seed=2).to(dev), transpose=False)
    with pytest.raises(L.CrdrHipError):
        ops.conv2d_raw(x, w7, 32, (7, 7), 1, 3, False, (8, 8), algo=_wino_id())
    w3 = ops.pack_weight(_rand(32, 32, 3, 3, seed=2).to(dev), transpose=False)
    with pytest.raises(L.CrdrHipError):
        ops.conv2d_raw(x, w3, 32, (3, 3), 2, 1, False, (4, 4), algo=_wino_id())
    with pytest.raises(L.CrdrHipError):   # F(4x4, 3x3): fewer than 24 output columns
        ops.conv2d_raw(x, w3, 32, (3, 3), 1, 1, False, (8, 8), algo=_wino_id() + 2)
    with pytest.raises(L.CrdrHipError):   # the 5x5 stride-1 form wants >= 12 input channels
        w5 = ops.pack_weight(_rand(32, 8, 5, 5, seed=2).to(dev), transpose=False)
        ops.conv2d_raw(_rand(1, 8, 8, 64, seed=1).to(dev), w5, 32, (5, 5), 1, 2, False, (8, 64), algo=_wino_id() + 2)


WG_CASES = [
    # name, N, Cin, H, W, Cout, pad, log2 split
    ("96_96_16", 2, 96, 16, 16, 96, 1, 0),
    ("128_128_ragged", 1, 128, 34, 38, 128, 1, 2),
    ("64_160_odd", 2, 64, 15, 15, 160, 1, 1),
    ("36_40_c4", 1, 36, 20, 20, 40, 1, 0),
    ("32_64_valid", 1, 32, 18, 21, 64, 0, 3),
    ("8_8_tiny", 1, 8, 5, 3, 8, 1, 0),
]


@pytest.mark.parametrize("variant", [0, 1], ids=["f3x3_2x2", "f3x3_4x4"])
@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_winograd_wgrad(case, variant):
    """weight gradient through the Winograd slab kernels -- F(3x3, 2x2) (the last forced wgrad configuration) and F(3x3, 4x4) (the id
    behind it, wino4_wgrad.hip) -- vs fp64 torch, for the Conv2d and the ConvTranspose2d operand order, with and without accumulation"""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, p, ls = case
    dev = _dev()
    if variant and (n * ((h - 2 + 2 * p + 3) // 4) * ((w - 2 + 2 * p + 15) // 16)) >> ls < 1:
        ls = 0   # (fewer strips of 4 x 16 pixels than the split asks for)
    algo = (L.load().crdr_conv2d_wgrad_num_configs() + variant) | (ls << 8)
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, 3, 3, seed=2, scale=(ci * 9) ** -0.5)
    xr, wr = x.double(), wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=p)
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    g = torch.zeros(co, ci, 3, 3, device=dev)
    ops.conv2d_wgrad_raw(dyd, xd, g, (3, 3), 1, p, accumulate=False, algo=algo)
    _close(g, wr.grad, name + " wgrad")
    gd = torch.zeros_like(g)
    ops.conv2d_wgrad_raw(dyd, xd, gd, (3, 3), 1, p, accumulate=False, algo=1)
    e_w = (g.cpu().double() - wr.grad).abs().max().item()
    e_d = (gd.cpu().double() - wr.grad).abs().max().item()
    print(f"{name}: wgrad max err winograd {e_w:.2e} direct {e_d:.2e} (scale {wr.grad.abs().max().item():.2e})")
    ops.conv2d_wgrad_raw(dyd, xd, g, (3, 3), 1, p, accumulate=True, algo=algo)
    _close(g, 2 * wr.grad, name + " wgrad accumulate")
    if p == 1:   # ConvTranspose2d 3x3 s1: P = x, Q = dy
        wT = _rand(ci, co, 3, 3, seed=5, scale=(ci * 9) ** -0.5).double().requires_grad_(True)
        rT = F.conv_transpose2d(xr, wT, None, padding=1)
        dyT = _rand(*rT.shape, seed=6)
        rT.backward(dyT.double())
        gT = torch.zeros(ci, co, 3, 3, device=dev)
        ops.conv2d_wgrad_raw(xd, dyT.to(dev), gT, (3, 3), 1, 1, accumulate=False, algo=algo)
        _close(gT, wT.grad, name + " convT wgrad")
