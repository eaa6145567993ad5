"""Why the gradients UPSTREAM of the quantiser (encoder.*, hyperencoder.*) are held to a looser bound than everything behind it -- the
argument of tests/test_gpu_step.py (UPSTREAM_TUNED_TOL), tested on the CPU oracle alone, no GPU.  The claim under test:

  forward activations of two correct fp32 implementations differ by summation-order noise (~1e-6 of a layer's output scale for the
  direct kernels, ~1.5e-6 / ~6e-6 for the F(4x4) kernels at the round-5 / round-4 points); a pre-activation inside that noise band around
  zero flips its ReLU mask in the backward pass, and every flipped element adds a FINITE rank-one term to the weight gradients above it.
  Behind the quantiser the rounding absorbs the noise (same symbols, same y_hat): nothing comparable happens there.

The oracle's own stage-1 generator step (rate_distortion_trainer.py:57-75) is run in float64 with relative Gaussian noise of a chosen size
injected behind every convolution of the analysis and hyper-analysis transforms, rounding decisions shared with the clean run:

 (1) plain fp32 rounding (the oracle in float32 against itself in float64) flips no mask at this size and is 1e-6-class in EVERY module:
     the loose upstream bound is not about precision as such;
 (2) noise of the kernels' size reproduces what the GPU measured -- 1e-6 -> 1.2e-3 upstream (GPU, direct / F(2x2) plans: 1.2e-3 on the
     hyper-encoder), 6e-6 -> 3.3e-3 (GPU, F(4x4) at the round-4 points: 2.6e-3 .. 9.7e-3; profiles/r4_parity_margins.json) -- with a
     handful of flipped masks (1 / 10 / 55 of 5.5 M at 2.5e-7 / 1e-6 / 6e-6) and 25 .. 50 times less behind the quantiser;
 (3) with the clean run's masks IMPOSED the same noise moves the upstream gradients 25x less: the flips carry it.
An earlier version of the argument (round 4) also blamed the heavy-tailed -1 / (p ln 2) rate gradient for the size of the jumps; (3) runs
the comparison with the rate term switched off and finds the same discrepancy, so that part is withdrawn: a single flipped element of a
late, small layer is a 1e-4-class change of the encoder's gradient whichever loss term drives it."""
import os

import pytest
import torch

from tests.golden.seeded_weights import seeded_input, seeded_tensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UP = ("encoder", "hyperencoder")


def _shapes():
    """state-dict shapes of the stage-1 generator, from the product's own module tree (built on the CPU: no launch happens)"""
    from crdr_amd.models import build_comp_model
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", "_base_", "model", "elic_charm.yaml"))
    cfg["device"] = "cpu"
    model = build_comp_model(ConfigDict(cfg))
    return {k: tuple(p.shape) for k, p in model.named_parameters()}


def _run(shapes, dtype, noise_rel=0.0, symbols=None, rate_on=True, masks=None, record=None, impose=None):
    """one stage-1 generator step of the oracle.  `record`: list that receives every rounding decision in call order; `symbols`: such a list
    from another run -- its decisions are taken over one for one (the quantiser then absorbs whatever noise reaches it, exactly as it does
    between two correct implementations whose decisions agree: tests/test_gpu_step.py gates that with oracle.check_forced); `masks`: list
    that receives every ReLU's mask; `impose`: such a list from another run -- its masks are applied instead of this run's own"""
    from oracle import crdr_oracle as O
    sd = {k: seeded_tensor(k, s).to(dtype).requires_grad_(True) for k, s in shapes.items()}
    x = seeded_input("image", (2, 3, 64, 64)).to(dtype)
    ny, nz = seeded_input("noise.y", (2, 320, 4, 4), 0.5).to(dtype), seeded_input("noise.z", (2, 192, 1, 1), 0.5).to(dtype)
    conv0, relu0, round0 = O.conv, O.F.relu, O.forced_round
    gen = torch.Generator().manual_seed(99)
    taken = iter(symbols) if symbols is not None else None
    imposed = iter(impose) if impose is not None else None

    def noisy_conv(sd_, name, x_, stride=1, pad=0):
        y = conv0(sd_, name, x_, stride=stride, pad=pad)
        if noise_rel and name.split(".")[0] in UP:   # summation-order noise of a correct implementation: relative to the output scale
            y = y + noise_rel * y.detach().abs().max() * torch.randn(y.shape, generator=gen, dtype=torch.float64).to(y.dtype)
        return y

    def spy_relu(t, *a, **k):
        if masks is not None:
            masks.append((t.detach() > 0))
        if imposed is not None:
            return t * next(imposed).to(t.dtype)
        return relu0(t, *a, **k)

    def rounding(v, forced=None, report=None):
        q = next(taken).to(v.dtype) if taken is not None else torch.round(v.detach())
        if record is not None:
            record.append(q.detach().clone())
        return q
    O.conv, O.F.relu, O.forced_round = noisy_conv, spy_relu, rounding
    try:
        out = O.generator_forward(sd, x, None, None, ny, nz)
        loss = O.mse_loss(x, out["fake_images"], 150.0)
        if rate_on:
            loss = loss + out["bpp"].mean()   # (lambda 1: HificRateLoss' weights are 2^-4 .. 2^1 over the stages, rate_loss.py:84-106)
        loss.backward()
    finally:
        O.conv, O.F.relu, O.forced_round = conv0, relu0, round0
    grads = {k: v.grad.detach().double() for k, v in sd.items() if v.grad is not None}
    return grads, out


def _by_module(a, b):
    acc = {}
    for k in b:
        top = k.split(".")[0]
        d, n = acc.setdefault(top, [0.0, 0.0])
        acc[top] = [d + float((a[k] - b[k]).square().sum()), n + float(b[k].square().sum())]
    return {k: (d / max(n, 1e-300)) ** 0.5 for k, (d, n) in acc.items()}


@pytest.fixture(scope="module")
def reference():
    shapes = _shapes()
    masks, symbols = [], []
    g64, _ = _run(shapes, torch.float64, masks=masks, record=symbols)
    return shapes, g64, masks, symbols


def test_fp32_rounding_alone_is_harmless_everywhere(reference):
    """the oracle in float32 against itself in float64, same rounding decisions: 1e-6-class in EVERY module, upstream of the quantiser
    included -- plain fp32 rounding (6e-8) flips next to no ReLU mask at this size.  The loose upstream bound is not about precision as such."""
    shapes, g64, masks0, symbols = reference
    masks = []
    g32, _ = _run(shapes, torch.float32, symbols=symbols, masks=masks)
    e = _by_module(g32, g64)
    flips = sum(int((a != b).sum()) for a, b in zip(masks0, masks))
    print("fp32 vs fp64 gradient, relative L2 per module:", {k: f"{v:.2e}" for k, v in e.items()}, "flipped masks:", flips)
    assert max(e.values()) < 2e-5, e


def test_upstream_discrepancy_follows_the_forward_noise_through_relu_mask_flips(reference):
    shapes, g64, masks0, symbols = reference
    res = {}
    for eps in (2.5e-7, 1e-6, 6e-6):
        masks = []
        g, _ = _run(shapes, torch.float64, noise_rel=eps, symbols=symbols, masks=masks)
        flips = sum(int((a != b).sum()) for a, b in zip(masks0, masks))
        total = sum(a.numel() for a in masks0)
        e = _by_module(g, g64)
        res[eps] = (max(e[k] for k in UP), max(v for k, v in e.items() if k not in UP), flips, total)
        print(f"noise {eps:.1e}: upstream {res[eps][0]:.2e}  downstream {res[eps][1]:.2e}  flipped ReLU masks {flips} of {total}")
    (u0, d0, f0, _), (u1, d1, f1, _), (u6, d6, f6, _) = res[2.5e-7], res[1e-6], res[6e-6]
    assert 0 < f1 and 2.0 * f1 < f6 < 18.0 * f1, (f0, f1, f6)          # the band around zero is 6x wider: ~6x the flips
    assert 6 ** 0.5 * 0.5 < u6 / u1 < 6 * 2.0, (u1, u6)                # between independent flips (square root) and linear growth
    assert u0 < u1 < u6, (u0, u1, u6)
    # the order the GPU measured at these noise levels of the analysis transform (direct / F(2x2) kernels ~1e-6: 1.2e-3 on the hyper-encoder;
    # F(4x4) at the round-4 points ~6e-6: 2.6e-3 .. 9.7e-3; round 5's points sit at ~1.5e-6)
    assert 2e-4 < u1 < 8e-3 and 6e-4 < u6 < 3e-2, (u1, u6)
    assert d6 < 0.1 * u6 and d1 < 0.1 * u1, (d1, u1, d6, u6)           # behind the quantiser the same noise is absorbed by the rounding


def test_the_flipped_masks_carry_it_not_the_noise_itself(reference):
    """the same forward noise with the reference run's ReLU masks imposed: the upstream discrepancy collapses (what is left is the smooth
    sensitivity of the sigmoid gates and of the weight-gradient operands); and it does not come from the rate term -- the handful of
    flipped elements sit in layers whose gradients the distortion term dominates just as well"""
    shapes, g64, masks0, symbols = reference
    g_free, _ = _run(shapes, torch.float64, noise_rel=6e-6, symbols=symbols)
    g_imp, _ = _run(shapes, torch.float64, noise_rel=6e-6, symbols=symbols, impose=masks0)
    e_free, e_imp = max(_by_module(g_free, g64)[k] for k in UP), max(_by_module(g_imp, g64)[k] for k in UP)
    print(f"upstream discrepancy at noise 6e-6: own masks {e_free:.2e}, reference masks imposed {e_imp:.2e}")
    assert e_imp < 0.1 * e_free, (e_imp, e_free)
    g0, _ = _run(shapes, torch.float64, symbols=symbols, rate_on=False)
    gn, _ = _run(shapes, torch.float64, noise_rel=6e-6, symbols=symbols, rate_on=False)
    e_off = max(_by_module(gn, g0)[k] for k in UP)
    print(f"upstream discrepancy at noise 6e-6 with the rate term off: {e_off:.2e}")
    assert 0.3 * e_free < e_off < 3.0 * e_free, (e_off, e_free)


def test_imposed_masks_are_adopted_only_inside_the_window(reference):
    """oracle.relu / generator_forward(impose=...) -- the mechanism behind the deterministic upstream-gradient gate of tests/test_gpu_step.py.
    (a) the oracle's OWN masks imposed change nothing and adopt nothing; (b) a mask that disagrees where the pre-activation is far from zero is
    counted as `outside` and fails check_imposed; (c) a disagreement inside the window is adopted and moves the gradients."""
    from oracle import crdr_oracle as O
    shapes = reference[0]
    sd = {k: seeded_tensor(k, s).double().requires_grad_(True) for k, s in shapes.items()}
    x = seeded_input("image", (1, 3, 64, 64)).double()
    ny, nz = seeded_input("noise.y", (1, 320, 4, 4), 0.5).double(), seeded_input("noise.z", (1, 192, 1, 1), 0.5).double()
    own, pre = {}, {}
    relu0 = O.relu

    def spy(t, site):
        if t.requires_grad:
            own[site], pre[site] = (t.detach() > 0), t.detach().clone()
        return relu0(t, site)
    O.relu = spy
    try:
        out0 = O.generator_forward(sd, x, None, None, ny, nz)
    finally:
        O.relu = relu0
    assert len(own) == 150, len(own)
    (out0["fake_images"].square().mean() + out0["bpp"].mean()).backward()
    g0 = {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}

    def run(masks):
        for v in sd.values():
            v.grad = None
        imp = {"masks": masks, "report": {}}
        out = O.generator_forward(sd, x, None, None, ny, nz, impose=imp)
        (out["fake_images"].square().mean() + out["bpp"].mean()).backward()
        return {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}, imp["report"]
    g1, rep = run(own)
    O.check_imposed(rep)
    assert rep["sites"] == 150 and rep.get("flipped", 0) == 0 and all(torch.equal(g0[k], g1[k]) for k in g0), rep
    # (b) far from zero
    site = "encoder.block1.block0.conv.0"
    far = {k: v.clone() for k, v in own.items()}
    idx = pre[site].abs().flatten().argmax()
    far[site].view(-1)[idx] = ~far[site].view(-1)[idx]
    _, rep = run(far)
    assert rep["flipped"] >= 1 and rep["outside"] >= 1, rep   # (the forward goes on from the wrong activation: later sites disagree as well)
    with pytest.raises(AssertionError):
        O.check_imposed(rep)
    # (c) inside the window: the element closest to zero
    near = {k: v.clone() for k, v in own.items()}
    idx = pre[site].abs().flatten().argmin()
    assert float(pre[site].abs().flatten()[idx]) <= O.MASK_WINDOW * float(pre[site].abs().max())
    near[site].view(-1)[idx] = ~near[site].view(-1)[idx]
    g2, rep = run(near)
    O.check_imposed(rep)
    assert 1 <= rep["flipped"] <= 4 and rep.get("outside", 0) == 0, rep
    assert any(not torch.equal(g0[k], g2[k]) for k in g0 if k.startswith("encoder.conv1"))
