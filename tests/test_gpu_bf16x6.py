"""Opt-in fp32-EQUIVALENT matrix mode `precision: bf16x6` (include/crdr_hip.h CRDR_CONV_BF16X6 / CRDR_WGRAD_BF16X6).

Every operand is split EXACTLY into three bf16 pieces (hi + mid + lo = x: 3 x 8 bits of the 24-bit significand) and a product is the six piece
products of weight >= 2^-16 (am bm, al bh, ah bl, am bh, ah bm, ah bh) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, small terms first.
Dropped: am bl, al bm, al bl <= 2^-23 |a b| together -- one fp32 rounding, what the exact-fp32 MFMA chain pays per accumulation step anyway.

Stated tolerance = the fp32 gates, UNCHANGED: kernels against float64 at the direct fp32 kernels' own gate (2e-5 of the output scale in
tests/test_gpu_conv.py; measured here and held to 4e-6, the worst exact-fp32 direct entry of the plan replay), elementwise within 2^-21 sum |a||b|,
the full-size training steps against the oracle at the gates of tests/test_gpu_step.py (same functions, `precision=` argument).  The codec
never runs this mode (byte parity stays on the exact fp32 instruction)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_conv import _dev, _rand

pytestmark = pytest.mark.gpu
ELEM_BOUND = 2.0 ** -21      # x sum |a||b| (+ 1e-7 of the output scale for fp32 accumulation of near-cancelling sums)
SCALE_GATE = 4e-6            # max error / max |reference|: the exact-fp32 direct kernels' own worst in profiles/r5_plan_replay.json


@pytest.fixture()
def bf16x6():
    from crdr_amd.hip import ops
    ops.MATRIX_BF16X6 = True
    yield
    ops.MATRIX_BF16X6 = False


CASES = [("3x3_96_96", 2, 96, 16, 16, 96, 3, 1, 1), ("5x5s2_192_320", 2, 192, 16, 16, 320, 5, 2, 2), ("5x5_224_128", 4, 224, 8, 8, 128, 5, 1, 2),
         ("1x1_320_160", 1, 320, 6, 10, 160, 1, 1, 0), ("3x3s2_64_64", 2, 64, 18, 22, 64, 3, 2, 1), ("1x1_256_128", 4, 256, 64, 64, 128, 1, 1, 0),
         ("3x3_128_128", 2, 128, 32, 32, 128, 3, 1, 1), ("5x5_352_224", 3, 352, 8, 8, 224, 5, 1, 2)]


def _refs(case):
    name, n, ci, h, w, co, k, s, p = case
    x = _rand(n, ci, h, w, seed=1)
    wt = _rand(co, ci, k, k, seed=2, scale=(ci * k * k) ** -0.5)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = _rand(*ref.shape, seed=4)
    ref.backward(dy.double())
    xa, wa, da = x.double().abs(), wt.double().abs(), dy.double().abs()
    b_fwd = F.conv2d(xa, wa, None, stride=s, padding=p)
    b_dx = torch.autograd.grad(F.conv2d(xa.requires_grad_(True), wa, None, stride=s, padding=p), xa, da)[0]
    b_dw = torch.autograd.grad(F.conv2d(xa.detach(), wa.requires_grad_(True), None, stride=s, padding=p), wa, da)[0]
    return x, wt, dy, (ref.detach(), xr.grad, wr.grad), (b_fwd, b_dx.detach(), b_dw.detach())


def _check(got, want, bound, what):
    err = (got.detach().cpu().double() - want).abs()
    scale = float(want.abs().max())
    worst = float((err / (ELEM_BOUND * bound + 1e-7 * scale)).max())
    rel = float(err.max()) / scale
    assert worst <= 1.0 and rel <= SCALE_GATE, f"{what}: {worst:.2f} x the elementwise bound, {rel:.2e} of the output scale (gate {SCALE_GATE:.0e})"
    return rel


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_bf16x6_conv_family_at_the_fp32_gate(case, bf16x6):
    """Built-in plans of forward, input gradient and weight gradient in the mode against float64; the same launches on the exact fp32
    instruction beside them: the mode's error is of the exact kernels' size (<= 3x, or 1.5e-6 of the output scale: the plans -- hence the summation orders -- of the two modes differ), not of bf16x3's (1e-5)."""
    from crdr_amd.hip import ops
    name, n, ci, h, w, co, k, s, p = case
    dev = _dev()
    x, wt, dy, refs, bounds = _refs(case)
    oh, ow = refs[0].shape[2:]
    xd, wd, dyd = x.to(dev), wt.to(dev), dy.to(dev)

    def run():
        out = ops.conv2d_raw(xd, ops.pack_weight(wd, False), co, (k, k), s, p, False, (oh, ow))
        dx = ops.conv2d_raw(dyd, ops.pack_weight(wd, True), ci, (k, k), s, p, True, (h, w))
        g = torch.zeros_like(wd)
        ops.conv2d_wgrad_raw(dyd, xd, g, (k, k), s, p, accumulate=False, defer=False)
        torch.cuda.synchronize()
        return out, dx, g
    got6 = run()
    assert ops.MATRIX_BF16X6
    ops.MATRIX_BF16X6 = False
    got32 = run()
    ops.MATRIX_BF16X6 = True
    for a6, a32, want, bound, what in zip(got6, got32, refs, bounds, ("fwd", "dgrad", "wgrad")):
        r6 = _check(a6, want, bound, f"{name} {what}")
        r32 = float((a32.detach().cpu().double() - want).abs().max()) / float(want.abs().max())
        assert r6 <= max(3.0 * r32, 1.5e-6), f"{name} {what}: bf16x6 {r6:.2e} vs exact fp32 {r32:.2e} of the output scale"
        assert not torch.equal(a6, a32), f"{name} {what}: the mode did not run (bit-identical to the exact path)"
        print(f"{name:16s} {what:6s} bf16x6 {r6:.2e}  exact fp32 {r32:.2e}  (of the output scale, vs float64)")


def test_bf16x6_every_tile_configuration_and_split(bf16x6):
    """Every forced tile configuration x split depth, every streaming 1x1 variant and every direct weight-gradient configuration in the mode."""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    dev = _dev()
    lib = L.load()
    ran = {"conv": 0, "stream": 0, "wgrad": 0}
    for case in (("3x3_160_160", 16, 160, 16, 16, 160, 3, 1, 1), ("1x1_192_96", 2, 192, 96, 96, 96, 1, 1, 0)):
        name, n, ci, h, w, co, k, s, p = case
        x, wt, dy, refs, bounds = _refs(case)
        oh, ow = refs[0].shape[2:]
        xd, wd, dyd = x.to(dev), wt.to(dev), dy.to(dev)
        wp = ops.pack_weight(wd, False)
        algos = [(c + 1) | (ls << 8) for c in range(lib.crdr_conv2d_num_configs()) for ls in range(4)]
        nc = lib.crdr_conv2d_num_configs()
        algos += [nc + 1 + v for v in range(lib.crdr_conv2d_num_stream_configs())] if k == 1 else []
        for algo in algos:
            try:
                out = ops.conv2d_raw(xd, wp, co, (k, k), s, p, False, (oh, ow), algo=algo)
            except L.CrdrHipError:
                continue
            _check(out, refs[0], bounds[0], f"{name} fwd algo {algo:#x}")
            ran["stream" if (algo & 0xff) > nc else "conv"] += 1
        for c in range(lib.crdr_conv2d_wgrad_num_configs() - 1):
            for ls in (0, 2):
                g = torch.zeros_like(wd)
                try:
                    ops.conv2d_wgrad_raw(dyd, xd, g, (k, k), s, p, accumulate=False, defer=False, algo=(c + 1) | (ls << 8))
                except L.CrdrHipError:
                    continue
                _check(g, refs[2], bounds[2], f"{name} wgrad cfg {c} split {1 << ls}")
                ran["wgrad"] += 1
    assert ran["conv"] >= 20 and ran["stream"] >= 4 and ran["wgrad"] >= 40, ran


def test_bf16x6_epilogues_and_grouped_launches_match_the_exact_path(bf16x6):
    """The mode changes the products only: a launch with the fused epilogue (bias, ReLU, residual, affine) and a grouped launch agree with the
    exact-fp32 launch of the same plan to the mode's accuracy."""
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    dev = _dev()
    x = _rand(2, 128, 24, 24, seed=1).to(dev)
    wt = _rand(128, 128, 3, 3, seed=2, scale=0.03).to(dev)
    b, sc, sh = _rand(128, seed=3).to(dev), (_rand(128, seed=5) * 0.5 + 1).to(dev), _rand(128, seed=6).to(dev)
    res = _rand(2, 128, 24, 24, seed=7).to(dev).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_weight(wt, False)
    fl = L.EPI_BIAS | L.EPI_RELU | L.EPI_RES | L.EPI_AFFINE

    def run():
        return ops.conv2d_raw(x, wp, 128, (3, 3), 1, 1, False, (24, 24), bias=b, res=res, scale=sc, shift=sh, flags=fl)
    a6 = run()
    ops.MATRIX_BF16X6 = False
    a32 = run()
    ops.MATRIX_BF16X6 = True
    d = float((a6 - a32).abs().max()) / float(a32.abs().max())
    assert 0.0 < d <= 2e-6, d


def test_bf16x6_mode_is_opt_in_and_exclusive():
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    assert ops.MATRIX_BF16X6 is False and ops.MATRIX_BF16X3 is False
    dev = _dev()
    x = _rand(1, 32, 8, 8, seed=1).to(dev)
    wp = ops.pack_weight(_rand(32, 32, 3, 3, seed=2).to(dev), False)
    with pytest.raises(L.CrdrHipError, match="exclusive"):
        ops.conv2d_raw(x, wp, 32, (3, 3), 1, 1, False, (8, 8), flags=L.CONV_BF16X3 | L.CONV_BF16X6)


def test_bf16x6_stage3_step_at_the_fp32_gates():
    """One stage-3 step with `precision: bf16x6` through tests/test_gpu_step.py's own comparison (same function, same gates as the exact-fp32
    test: losses 3e-4, gradients 5e-3, the analysis / hyper-analysis transforms under the imposed-mask gate of 5e-4) at 64 x 64 on built-in plans,
    and the mode does not leak out of the step."""
    from crdr_amd.hip import ops
    from tests.test_gpu_step import _stage3_step
    _stage3_step(precision="bf16x6", impose_masks=True)
    assert ops.MATRIX_BF16X6 is False and ops.MATRIX_BF16X3 is False, "the mode must not leak out of the training step"


def test_bf16x6_stage3_step_256_tuned_vs_oracle():
    """BASELINE config #3 (bs 16, 256 x 256) with `precision: bf16x6` on the shipped plan set of that mode (the database's bf16x6 entries: the
    tuner's choice between the split-bf16 direct kernels and the exact-fp32 Winograd kernels per shape) against the oracle at the UNCHANGED
    fp32 gates of test_stage3_step_256_tuned_vs_oracle."""
    from tests import parity_margins as PM
    from tests.test_gpu_step import _ShippedPlans, _stage3_step
    with _ShippedPlans() as sp:
        _stage3_step(bs=16, size=256, precision="bf16x6", impose_masks=True)
        new = sp.tuned_here()
    PM.record("plans", "shapes tuned on the spot (not in the shipped database)", float(len(new)))
    assert len(new) <= 8, new


def test_bf16x6_stage1_step_256_tuned_vs_oracle():
    """BASELINE config #2 (bs 8, 256 x 256) likewise"""
    from tests import parity_margins as PM
    from tests.test_gpu_step import _ShippedPlans, _stage1_step
    with _ShippedPlans() as sp:
        _stage1_step(bs=8, size=256, precision="bf16x6", impose_masks=True)
        new = sp.tuned_here()
    PM.record("plans", "shapes tuned on the spot (not in the shipped database)", float(len(new)))
    assert len(new) <= 8, new
