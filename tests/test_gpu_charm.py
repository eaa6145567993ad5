"""GPU parity of the fused Charm engine (crdr_amd/hip/charm.py) and of the kernel features it is built from, through
the C ABI: pre-activation addend / ReLU-mask epilogues, grouped conv and weight-gradient launches, sub-block weight
packs, scattered weight-gradient reductions, scattered column sums, in-kernel Philox noise.

Checker: the CPU oracle's `charm_forward` (oracle/crdr_oracle.py, restating minnen20_charm_context_model.py:88-141 and
pinned by tests/golden `charm.y_hat / charm.lik`) on the same seeded weights; plain torch fp64 for the primitives."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.golden.seeded_weights import seeded_input, seeded_tensor
from tests.test_gpu_model import check_grads, close, dev, grad_close, grad_sd, rel, seed_module

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.to(dev()).contiguous(memory_format=torch.channels_last)


def wide(m, c, fill=None):
    b = torch.empty((m, c), dtype=torch.float32, device=dev())
    if fill is not None:
        b.fill_(fill)
    return b


def test_preadd_mask_grouped_conv():
    """G = 3 problems, 5x5 'same' conv 64 -> 96 reading / writing channel ranges of wide buffers:
    out = relu(conv(x_g, w_g) + pre_g + b_g), then a grouped transposed launch with ReLU mask and ACCUM."""
    from crdr_amd.hip import functional as HF, lib as L, ops
    from crdr_amd.hip.ops import V, view
    n, h, w, ci, co, G = 2, 9, 7, 64, 96, 3
    M = n * h * w
    xs = [seeded_input(f"cg.x{g}", (n, ci, h, w)) for g in range(G)]
    ws = [seeded_tensor(f"cg.w{g}.weight", (co, ci, 5, 5)) for g in range(G)]
    bs = [seeded_tensor(f"cg.b{g}.bias", (co,)) for g in range(G)]
    pre = [seeded_input(f"cg.pre{g}", (n, co, h, w)) for g in range(G)]
    X = wide(M, G * ci)
    Y = wide(M, G * co + 32, 7.0)
    for g in range(G):
        X.view(n, h, w, -1)[..., g * ci:(g + 1) * ci] = xs[g].permute(0, 2, 3, 1).to(dev())
        Y.view(n, h, w, -1)[..., g * co:(g + 1) * co] = pre[g].permute(0, 2, 3, 1).to(dev())
    packs = [ops.pack_weight(wt.to(dev()), False) for wt in ws]
    bd = [b.to(dev()) for b in bs]
    yv = [view(Y, g * co, co) for g in range(G)]
    ops.conv_group(n, h, w, [view(X, g * ci, ci) for g in range(G)], [p.data_ptr() for p in packs], yv, co, (5, 5), 2, False,
                   wrows=packs[0].shape[1], wcols=packs[0].shape[2], biases=[b.data_ptr() for b in bd], pres=yv, flags=L.EPI_RELU,
                   device=dev())
    got = Y.view(n, h, w, -1)
    for g in range(G):
        ref = F.relu(F.conv2d(xs[g].double(), ws[g].double(), bs[g].double(), padding=2) + pre[g].double())
        close(got[..., g * co:(g + 1) * co].permute(0, 3, 1, 2), ref, f"grouped preadd conv {g}", 2e-5)
    assert torch.all(got[..., G * co:] == 7.0), "the grouped launch wrote outside its channel ranges"
    # transposed (input-gradient) launch: dx_g += conv_T(dy_g, w_g) * (mask_g > 0)
    dys = [seeded_input(f"cg.dy{g}", (n, co, h, w)) for g in range(G)]
    masks = [seeded_input(f"cg.m{g}", (n, ci, h, w)) for g in range(G)]
    old = [seeded_input(f"cg.old{g}", (n, ci, h, w)) for g in range(G)]
    DY, MK, DX = wide(M, G * co), wide(M, G * ci), wide(M, G * ci)
    for g in range(G):
        DY.view(n, h, w, -1)[..., g * co:(g + 1) * co] = dys[g].permute(0, 2, 3, 1).to(dev())
        MK.view(n, h, w, -1)[..., g * ci:(g + 1) * ci] = masks[g].permute(0, 2, 3, 1).to(dev())
        DX.view(n, h, w, -1)[..., g * ci:(g + 1) * ci] = old[g].permute(0, 2, 3, 1).to(dev())
    dpacks = [ops.pack_weight(wt.to(dev()), True) for wt in ws]
    ops.conv_group(n, h, w, [view(DY, g * co, co) for g in range(G)], [p.data_ptr() for p in dpacks],
                   [view(DX, g * ci, ci) for g in range(G)], ci, (5, 5), 2, True, wrows=dpacks[0].shape[1], wcols=dpacks[0].shape[2],
                   masks=[view(MK, g * ci, ci) for g in range(G)], flags=L.EPI_ACCUM, device=dev())
    for g in range(G):
        ref = F.conv_transpose2d(dys[g].double(), ws[g].double(), padding=2) * (masks[g].double() > 0) + old[g].double()
        close(DX.view(n, h, w, -1)[..., g * ci:(g + 1) * ci].permute(0, 3, 1, 2), ref, f"grouped masked dgrad {g}", 2e-5)


def test_sub_block_packs_and_hoisted_conv():
    """Three convs 64(+32k) -> 32 whose first 64 input channels are shared: the shared part as ONE wide conv over an
    N-concatenated sub-block pack, the rest per conv with the pre-activation addend == the full convs."""
    from crdr_amd.hip import functional as HF, lib as L, ops
    from crdr_amd.hip.ops import view
    n, h, w, hm, co = 1, 6, 8, 64, 32
    M = n * h * w
    extra = [32, 64, 96]
    ws = [seeded_tensor(f"sb.w{g}.weight", (co, hm + e, 3, 3)).to(dev()) for g, e in enumerate(extra)]
    x = seeded_input("sb.x", (n, hm + max(extra), h, w))
    X = wide(M, hm + max(extra))
    X.view(n, h, w, -1).copy_(x.permute(0, 2, 3, 1))
    hyp = torch.empty((9, 3 * co, hm), dtype=torch.float32, device=dev())
    ents = [HF.sub_pack(ws[g], 0, hm, hyp, g * co * hm, co, hm, False, dld=hm, tstride=3 * co * hm) for g in range(3)]
    sup = [torch.empty((9, co, e), dtype=torch.float32, device=dev()) for e in extra]
    ents += [HF.sub_pack(ws[g], hm, hm + e, sup[g], 0, co, e, False) for g, e in enumerate(extra)]
    HF.ensure_fresh(ents)
    A = wide(M, 3 * co)
    ops.conv_group(n, h, w, [view(X, 0, hm)], [hyp.data_ptr()], [view(A, 0, 3 * co)], 3 * co, (3, 3), 1, False, wrows=3 * co, wcols=hm,
                   device=dev())
    for g, e in enumerate(extra):
        ops.conv_group(n, h, w, [view(X, hm, e)], [sup[g].data_ptr()], [view(A, g * co, co)], co, (3, 3), 1, False, wrows=co, wcols=e,
                       pres=[view(A, g * co, co)], device=dev())
        ref = F.conv2d(x[:, :hm + e].double(), ws[g].cpu().double(), padding=1)
        close(A.view(n, h, w, -1)[..., g * co:(g + 1) * co].permute(0, 3, 1, 2), ref, f"hoisted + support conv {g}", 2e-5)
    # batched refill through a PackTable (what the fused Adam does) gives the same packs
    before = [hyp.clone()] + [s.clone() for s in sup]
    for b in [hyp] + sup:
        b.fill_(-1.0)
    flat_lo = min(wt.data_ptr() for wt in ws)
    flat_hi = max(wt.data_ptr() + wt.numel() * 4 for wt in ws)

    class _Flat:  # an address range standing in for an optimiser's flat parameter buffer
        device = dev()

        def data_ptr(self):
            return flat_lo

        def numel(self):
            return (flat_hi - flat_lo) // 4
    tb = HF.PackTable(_Flat())
    tb.refill()
    mine = {id(e) for e in ents}
    assert mine <= {id(e) for e in tb.entries}
    for a, b in zip(before, [hyp] + sup):
        assert torch.equal(a, b), "batched sub-block pack differs from the single-item pack"


def test_wgrad_split_and_grouped():
    """One slab launch feeding three parameters' input-channel ranges (crdr_wgrad_job.gJtot), and a grouped launch."""
    from crdr_amd.hip import ops
    from crdr_amd.hip.ops import view
    n, h, w, ci, co = 2, 8, 8, 64, 32
    M = n * h * w
    x = seeded_input("ws.x", (n, ci, h, w))
    dys = [seeded_input(f"ws.dy{g}", (n, co, h, w)) for g in range(3)]
    X, DY = wide(M, ci), wide(M, 3 * co)
    X.view(n, h, w, -1).copy_(x.permute(0, 2, 3, 1))
    for g in range(3):
        DY.view(n, h, w, -1)[..., g * co:(g + 1) * co] = dys[g].permute(0, 2, 3, 1).to(dev())
    tot = [ci + 32, ci + 64, ci]  # second dim of the three parameters; the job fills channels [j0, j0 + ci)
    j0 = [32, 0, 0]
    gs = [torch.full((co, t, 3, 3), 0.5, dtype=torch.float32, device=dev()) for t in tot]
    d = ops.DeferredWgrad(dev(), arena_bytes=64 << 20)
    prev, ops.WGRAD_DEFER = ops.WGRAD_DEFER, d
    try:
        ops.wgrad_split(n, h, w, view(DY, 0, 3 * co), view(X, 0, ci),
                        [(g * co, co, gs[g].data_ptr() + 4 * j0[g] * 9, tot[g]) for g in range(3)], (3, 3), 1, device=dev())
        d.flush("t1")
        for g in range(3):
            xr = x.double().requires_grad_(False)
            wr = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
            (F.conv2d(xr, wr, padding=1) * dys[g].double()).sum().backward()
            ref = torch.full((co, tot[g], 3, 3), 0.5, dtype=torch.float64)
            ref[:, j0[g]:j0[g] + ci] += wr.grad
            close(gs[g], ref, f"split wgrad {g}", 2e-5)
        # grouped: three (dy_g, x) -> g_g
        g2 = [torch.zeros((co, ci, 3, 3), dtype=torch.float32, device=dev()) for _ in range(3)]
        ops.wgrad_group(n, h, w, [view(DY, g * co, co) for g in range(3)], [view(X, 0, ci)] * 3, [(t.data_ptr(), 0) for t in g2], co, ci,
                        (3, 3), 1, device=dev())
        d.flush("t2")
        for g in range(3):
            wr = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
            (F.conv2d(x.double(), wr, padding=1) * dys[g].double()).sum().backward()
            close(g2[g], wr.grad, f"grouped wgrad {g}", 2e-5)
    finally:
        ops.WGRAD_DEFER = prev


def test_colsum_scatter():
    from crdr_amd.hip import ops
    from crdr_amd.hip.ops import view
    M, blk, nb = 300, 32, 5
    x = torch.randn(M, nb * blk + 16, generator=torch.Generator().manual_seed(3)).to(dev())
    outs = [torch.full((blk,), float(k), device=dev()) for k in range(nb)]
    table = torch.tensor([o.data_ptr() if k != 2 else 0 for k, o in enumerate(outs)], dtype=torch.int64, device=dev())
    ops.colsum_scatter(view(x, 0, nb * blk), M, blk, table, dev())
    for k in range(nb):
        ref = torch.full((blk,), float(k), dtype=torch.float64)
        if k != 2:
            ref += x[:, k * blk:(k + 1) * blk].double().sum(0).cpu()
        close(outs[k], ref, f"colsum block {k}", 1e-5)


def test_philox_noise_is_uniform_and_reproducible():
    from crdr_amd.hip import lib as L, ops
    lib = L.load()
    st = torch.tensor([1234, 0], dtype=torch.int64, device=dev())
    call = torch.empty(2, dtype=torch.int64, device=dev())
    n, hw, c = 4, 64, 320
    a, b, a2 = (torch.empty((n * hw, c), device=dev()) for _ in range(3))
    L.check(lib.crdr_philox_fork(st.data_ptr(), call.data_ptr(), 77, ops._stream()))
    L.check(lib.crdr_philox_uniform(call.data_ptr(), n, hw, c, c, 0, a.data_ptr(), c, ops._stream()))
    L.check(lib.crdr_philox_uniform(call.data_ptr(), n, hw, c, c, 0, a2.data_ptr(), c, ops._stream()))
    assert st.tolist() == [1234, 77] and call.tolist() == [1234, 0]
    L.check(lib.crdr_philox_uniform(st.data_ptr(), n, hw, c, c, 0, b.data_ptr(), c, ops._stream()))
    assert torch.equal(a, a2) and not torch.equal(a, b)
    assert a.min().item() >= -0.5 and a.max().item() < 0.5
    assert abs(a.mean().item()) < 5e-3 and abs(a.var().item() - 1 / 12) < 2e-3
    # a channel slice draws the same samples as the full tensor at those channels
    sl = torch.empty((n * hw, 32), device=dev())
    L.check(lib.crdr_philox_uniform(call.data_ptr(), n, hw, 32, c, 96, sl.data_ptr(), 32, ops._stream()))
    assert torch.equal(sl, a[:, 96:128])
    # lag-1 correlation along channels and pixels
    for v in (a[:, 1:] * a[:, :-1], a[1:] * a[:-1]):
        assert abs(v.mean().item() * 12) < 1e-2


def _charm_pair(max_support=5):
    from crdr_amd.models.subnet.context_model.minnen20_charm_context_model import Minnen20CharmContextModel
    from crdr_amd.models.subnet.entropy_model.ste_gaussian_conditional import SteGaussianMeanScaleConditional
    m = Minnen20CharmContextModel(num_slices=10, bottleneck_y=320, hyper_out_ch=640, max_support_slices=max_support)
    sd = seed_module(m, "context_model.")
    m.to(dev())
    return m, SteGaussianMeanScaleConditional().to(dev()), sd


@pytest.mark.parametrize("shape", [(2, 4, 4), (1, 5, 3)])
def test_charm_engine_matches_oracle(shape):
    """forward (y_hat, both likelihood tensors, both bit sums), gradients of y, hyper_out and every one of the 180
    parameters; rounding decisions must agree exactly outside FORCE_TOL of a boundary and the adopted ones are bounded."""
    from oracle import crdr_oracle as O
    n, h, w = shape
    m, em, sd = _charm_pair()
    y = seeded_input("charm.y", (n, 320, h, w), 4.0)
    hy = seeded_input("charm.hyper", (n, 640, h, w), 2.0)
    noise = seeded_input("charm.noise", (n, 320, h, w), 0.5)
    cot = seeded_input("charm.cot", (n, 320, h, w))
    gb = torch.linspace(0.5, 1.5, n)
    yd, hd = nhwc(y).requires_grad_(True), nhwc(hy).requires_grad_(True)
    m.record_symbols = []
    bits = {}
    yh, lik, qlik = m(yd, hd, em, is_train=True, noise=nhwc(noise), want_lik=True, bits_out=bits)
    forced = [t.cpu() for t in m.record_symbols]
    m.record_symbols = None
    sdg = grad_sd(sd)
    yg, hg = y.clone().requires_grad_(True), hy.clone().requires_grad_(True)
    rep = {}
    ryh, rlik, rqlik = O.charm_forward(sdg, yg, hg, noise, forced=forced, report=rep)
    O.check_forced(rep, rep.get("symbols", 0))
    close(yh, ryh, "charm y_hat", 2e-4)
    close(lik, rlik, "charm lik", 5e-4)
    close(qlik, rqlik, "charm qlik", 5e-4)
    close(bits["y"], O.bits_per_image(rlik), "charm bits", 2e-4)
    close(bits["y_q"], O.bits_per_image(rqlik), "charm qbits", 2e-4)
    ((yh * cot.to(dev())).sum() + (bits["y"] * gb.to(dev())).sum()).backward()
    ((ryh * cot).sum() + (O.bits_per_image(rlik) * gb).sum()).backward()
    grad_close(yd.grad, yg.grad, "charm dy")
    grad_close(hd.grad, hg.grad, "charm dhyper")
    check_grads(m, "context_model.", sdg, "charm")
    # eval mode (quantised likelihood only) and the reconstruction-only pass reproduce y_hat bit for bit
    with torch.no_grad():
        b2 = {}
        yh_e, lik_e, _ = m(nhwc(y), nhwc(hy), em, is_train=False, want_lik=True, bits_out=b2)
        yh_r = m.reconstruct_latent(nhwc(y), nhwc(hy)[:, :320], em)
    assert torch.equal(yh_e, yh.detach()) and torch.equal(yh_r, yh.detach())
    assert torch.equal(lik_e, qlik)
    assert torch.equal(b2["y"], bits["y_q"])


def test_charm_engine_all_slices_support():
    """max_support_slices = -1 (every decoded slice supports the later ones: no tail) against the oracle."""
    from oracle import crdr_oracle as O
    m, em, sd = _charm_pair(max_support=-1)
    n, h, w = 1, 4, 4
    y, hy = seeded_input("charm.y", (n, 320, h, w), 4.0), seeded_input("charm.hyper", (n, 640, h, w), 2.0)
    m.record_symbols = []
    with torch.no_grad():
        yh, lik, _ = m(nhwc(y), nhwc(hy), em, is_train=False, want_lik=True)
    rep = {}
    ryh, rlik, _ = O.charm_forward(sd, y, hy, None, max_support=10, forced=[t.cpu() for t in m.record_symbols], report=rep)
    O.check_forced(rep, rep.get("symbols", 0))
    close(yh, ryh, "charm(-1) y_hat", 2e-4)
    close(lik, rlik, "charm(-1) lik", 5e-4)


def test_charm_philox_training_noise():
    """Without an explicit noise tensor the kernels draw Philox noise: fresh per call, and the backward regenerates the
    forward's samples (gradients equal those of a run with the same samples passed explicitly)."""
    from crdr_amd.hip import lib as L, ops
    m, em, _ = _charm_pair()
    m.seed_noise(99)
    n, h, w = 2, 4, 4
    y, hy = seeded_input("charm.y", (n, 320, h, w), 4.0), seeded_input("charm.hyper", (n, 640, h, w), 2.0)
    st0 = m._philox(dev()).clone()
    lib = L.load()
    noise = torch.empty((n * h * w, 320), device=dev())
    L.check(lib.crdr_philox_uniform(st0.data_ptr(), n, h * w, 320, 320, 0, noise.data_ptr(), 320, ops._stream()))
    noise = noise.view(n, h, w, 320).permute(0, 3, 1, 2)

    def run(explicit):
        for p in m.parameters():
            p.grad = None
        yd, hd = nhwc(y).requires_grad_(True), nhwc(hy).requires_grad_(True)
        b = {}
        yh, _, _ = m(yd, hd, em, is_train=True, noise=explicit, want_lik=False, bits_out=b)
        (yh.square().mean() + b["y"].sum() * 1e-3).backward()
        return b["y"].clone(), yd.grad.clone(), hd.grad.clone()
    b1, gy1, gh1 = run(None)
    assert m._philox(dev()).tolist()[1] > 0
    b2, gy2, gh2 = run(noise)
    assert torch.equal(b1, b2) and torch.equal(gy1, gy2) and torch.equal(gh1, gh2)
    b3, _, _ = run(None)
    assert not torch.equal(b1, b3), "two training passes drew the same noise"
