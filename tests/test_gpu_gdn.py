"""GPU parity of the GDN / IGDN op (crdr_gdn_fwd / crdr_gdn_bwd through the C ABI) and of the Balle18 transforms built on
it, against the CPU oracle (oracle.gdn, restating compressai.layers.GDN as called from balle18_autoencoder.py:16-41)."""
import pytest
import torch

from tests.golden.seeded_weights import seeded_input
from tests.test_gpu_model import close, dev, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("c,n,h,w", [(192, 2, 9, 7), (64, 1, 16, 16)])
def test_gdn_fwd_bwd_matches_oracle(inverse, c, n, h, w):
    from oracle import crdr_oracle as O
    from crdr_amd.models.layer.gdn import GDN
    m = GDN(c, inverse=inverse)
    g = torch.Generator().manual_seed(c + int(inverse))
    with torch.no_grad():  # generic parameters: dense non-negative gamma, some entries stored BELOW their lower bounds
        m.gamma.copy_(torch.sqrt(torch.rand(c, c, generator=g) * 0.02 + O.GDN_REPARAM_OFFSET ** 2))
        m.gamma[0, :5] = 0.0
        m.beta.copy_(torch.sqrt(torch.rand(c, generator=g) + 0.5))
        m.beta[1] = 0.0
    sd = {"g.beta": m.beta.detach().clone().requires_grad_(True), "g.gamma": m.gamma.detach().clone().requires_grad_(True)}
    m.to(dev())
    x = seeded_input(f"gdn.x{c}", (n, c, h, w), 3.0)
    cot = seeded_input(f"gdn.cot{c}", (n, c, h, w))
    xr = x.clone().requires_grad_(True)
    ref = O.gdn(sd, "g", xr, inverse)
    (ref * cot).sum().backward()
    xd = x.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = m(xd)
    close(out, ref, "gdn forward", 2e-5)
    (out * cot.to(dev())).sum().backward()
    assert rel(xd.grad, xr.grad) < 2e-5, rel(xd.grad, xr.grad)
    assert rel(m.gamma.grad, sd["g.gamma"].grad) < 5e-5, rel(m.gamma.grad, sd["g.gamma"].grad)
    assert rel(m.beta.grad, sd["g.beta"].grad) < 5e-5, rel(m.beta.grad, sd["g.beta"].grad)
    # clamped entries follow the LowerBound rule exactly like the oracle (zero unless the gradient would raise them)
    assert torch.equal(m.gamma.grad[0, :5].cpu() == 0, sd["g.gamma"].grad[0, :5] == 0)


def test_balle18_transforms_build_and_run():
    from crdr_amd.utils.registry import DECODER_REGISTRY, ENCODER_REGISTRY
    import crdr_amd.models.subnet  # noqa: F401  (registration)
    torch.manual_seed(0)
    enc = ENCODER_REGISTRY.get("Balle18Encoder")(in_ch=3, out_ch=192, main_ch=192).to(dev())
    dec = DECODER_REGISTRY.get("Balle18Decoder")(in_ch=192, out_ch=3, main_ch=192, use_tanh=False).to(dev())
    x = seeded_input("image", (2, 3, 64, 64)).to(dev())
    y = enc(x)
    xh = dec(y)
    assert y.shape == (2, 192, 4, 4) and xh.shape == (2, 3, 64, 64)
    (xh - x).square().mean().backward()
    for p in list(enc.parameters()) + list(dec.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()
    keys = set(enc.state_dict())
    assert {"conv.0.weight", "conv.1.beta", "conv.1.gamma", "conv.1.beta_reparam.pedestal", "conv.1.gamma_reparam.lower_bound.bound",
            "conv.6.bias"} <= keys


@pytest.mark.parametrize("c", [192, 128])
def test_gdn_forward_is_stable_beside_a_loaded_chip(c):
    """The fused forward rewrites its x tile in place while the four waves of a workgroup read ALL channels of it: the waves must meet at a
    barrier between the matrix loop and the epilogue.  More tiles than CUs, a second stream keeping the chip unevenly busy (and, at 128
    channels, two workgroups per CU): every run must equal the float64 value of the op and the unloaded run bit for bit."""
    from crdr_amd.models.layer.gdn import GDN
    d = dev()
    m = GDN(c)
    g = torch.Generator().manual_seed(7 + c)
    with torch.no_grad():
        m.gamma.copy_(torch.sqrt(torch.rand(c, c, generator=g) * 0.02 + 2.0 ** -36))
        m.beta.copy_(torch.sqrt(torch.rand(c, generator=g) + 0.5))
    m.to(d)
    x = (torch.randn(8, c, 96, 96, generator=g) * 2.0).to(d).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y0 = m(x).clone()
        xd = x.double()
        ped = 2.0 ** -36
        gam = (torch.clamp(m.gamma.double(), min=2.0 ** -18) ** 2 - ped)
        bet = (torch.clamp(m.beta.double(), min=(1e-6 + ped) ** 0.5) ** 2 - ped)
        ref = xd / torch.sqrt(torch.einsum("ij,njhw->nihw", gam, xd * xd) + bet.view(1, -1, 1, 1))
        assert float((y0.double() - ref).abs().max() / ref.abs().max()) < 2e-5
        side = torch.cuda.Stream(device=d)
        a = torch.randn(4096, 4096, device=d)
        for rep in range(12):
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 3):
                    a = torch.tanh(a @ a * 1e-3)
            y = m(x)
            assert torch.equal(y, y0), f"run {rep} differs from the unloaded run"
        torch.cuda.synchronize()


def _gdn_f64(m, x, cot, inverse):
    """float64 value and gradients of the op on the device, from the formula (compressai's parametrisation with its LowerBound rule is
    covered by the oracle test above: here the parameters sit above their bounds)"""
    ped = 2.0 ** -36
    gp = m.gamma.detach().double().requires_grad_(True)
    bp = m.beta.detach().double().requires_grad_(True)
    gam = torch.clamp(gp, min=2.0 ** -18) ** 2 - ped
    bet = torch.clamp(bp, min=(1e-6 + ped) ** 0.5) ** 2 - ped
    xd = x.detach().double().requires_grad_(True)
    n = torch.einsum("ij,njhw->nihw", gam, xd * xd) + bet.view(1, -1, 1, 1)
    y = xd * torch.sqrt(n) if inverse else xd / torch.sqrt(n)
    (y * cot.double()).sum().backward()
    return y.detach(), xd.grad, bp.grad, gp.grad


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("c,n,h,w", [(192, 2, 100, 100), (128, 3, 90, 77), (100, 2, 100, 90), (64, 4, 75, 75), (192, 1, 5, 5), (160, 2, 70, 70),
                                     (36, 3, 80, 80)])
def test_gdn_fused_backward_many_tiles_vs_float64(inverse, c, n, h, w, monkeypatch):
    """The one-pass fused backward (gdn.hip: gdn_bwd_onepass_kernel, the CRDR_WGRAD_SQUARE_Q weight gradient, the per-workgroup column sums)
    with more 64-pixel tiles than workgroups, ragged last tiles and channel counts that do not fill the padded blocks: against float64 and
    against the nine-launch form (CRDR_GDN_UNFUSED_BWD=1)."""
    from crdr_amd.models.layer.gdn import GDN
    d = dev()
    m = GDN(c, inverse=inverse)
    g = torch.Generator().manual_seed(11 + c + int(inverse))
    with torch.no_grad():
        m.gamma.copy_(torch.sqrt(torch.rand(c, c, generator=g) * 0.02 + 2.0 ** -36))
        m.beta.copy_(torch.sqrt(torch.rand(c, generator=g) + 0.5))
    m.to(d)
    x = (torch.randn(n, c, h, w, generator=g) * 2.0).to(d).contiguous(memory_format=torch.channels_last)
    cot = torch.randn(n, c, h, w, generator=g).to(d).contiguous(memory_format=torch.channels_last)
    _, dx64, db64, dg64 = _gdn_f64(m, x, cot, inverse)

    def run():
        m.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        (m(xd) * cot).sum().backward()
        torch.cuda.synchronize()
        return xd.grad.clone(), m.beta.grad.clone(), m.gamma.grad.clone()
    fused = run()
    again = run()
    monkeypatch.setenv("CRDR_GDN_UNFUSED_BWD", "1")
    plain = run()
    monkeypatch.delenv("CRDR_GDN_UNFUSED_BWD")
    for name, a, b, r, tol in (("dx", fused[0], plain[0], dx64, 2e-5), ("dbeta", fused[1], plain[1], db64, 5e-5),
                                ("dgamma", fused[2], plain[2], dg64, 5e-5)):
        e_f, e_p = rel(a.double(), r), rel(b.double(), r)
        assert e_f < tol, (name, e_f, e_p)
        assert e_f < 3 * e_p + 2e-6, (name, e_f, e_p)   # no worse than the nine-launch form
    assert all(torch.equal(a, b) for a, b in zip(fused, again)), "the fused backward is not run-to-run identical"


def test_wgrad_square_q_equals_wgrad_of_the_squared_operand():
    """CRDR_WGRAD_SQUARE_Q (wgrad.hip): g = sum P Q^2 with the square taken inside the kernel -- the same bits as the launch on a squared
    copy (same configuration and split), and refused together with the split-bf16 forms."""
    from crdr_amd.hip import lib as L, ops
    d = dev()
    g_ = torch.Generator().manual_seed(5)
    for c in (192, 128, 64):
        p = torch.randn(3, c, 50, 41, generator=g_).to(d).contiguous(memory_format=torch.channels_last)
        q = torch.randn(3, c, 50, 41, generator=g_).to(d).contiguous(memory_format=torch.channels_last)
        q2 = (q * q).contiguous(memory_format=torch.channels_last)
        for cfg in range(L.load().crdr_conv2d_wgrad_num_configs()):
            for ls in (0, 3):
                a = (cfg + 1) | (ls << 8)
                g0, g1 = torch.empty(c, c, 1, 1, device=d), torch.empty(c, c, 1, 1, device=d)
                try:
                    ops.conv2d_wgrad_raw(p, q, g1, (1, 1), 1, 0, False, algo=a | L.WGRAD_SQUARE_Q)
                except L.CrdrHipError:
                    continue   # configurations without a squared form
                ops.conv2d_wgrad_raw(p, q2, g0, (1, 1), 1, 0, False, algo=a)
                assert torch.equal(g0, g1), (c, cfg, ls)
        g1 = torch.empty(c, c, 1, 1, device=d)
        ops.conv2d_wgrad_raw(p, q, g1, (1, 1), 1, 0, False, algo=L.WGRAD_SQUARE_Q)   # the plan's own choice among the squared forms
        ref = torch.einsum("nihw,njhw->ij", p.double(), q.double() ** 2)
        assert rel(g1.view(c, c).double(), ref) < 2e-5
        with pytest.raises(L.CrdrHipError):
            ops.conv2d_wgrad_raw(p, q, g1, (1, 1), 1, 0, False, algo=L.WGRAD_SQUARE_Q | L.WGRAD_BF16X6)


@pytest.mark.parametrize("c", [192, 128])
def test_gdn_backward_is_stable_beside_a_loaded_chip(c):
    """The one-pass backward hands dn from wave to wave through an LDS tile and reuses it and the x buffers from tile to tile: with more
    tiles than CUs and a second stream keeping the chip unevenly busy, every run must give the bits of the unloaded run."""
    from crdr_amd.models.layer.gdn import GDN
    d = dev()
    m = GDN(c)
    g = torch.Generator().manual_seed(17 + c)
    with torch.no_grad():
        m.gamma.copy_(torch.sqrt(torch.rand(c, c, generator=g) * 0.02 + 2.0 ** -36))
        m.beta.copy_(torch.sqrt(torch.rand(c, generator=g) + 0.5))
    m.to(d)
    x = (torch.randn(8, c, 96, 96, generator=g) * 2.0).to(d).contiguous(memory_format=torch.channels_last)
    cot = torch.randn(8, c, 96, 96, generator=g).to(d).contiguous(memory_format=torch.channels_last)

    def run():
        m.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        (m(xd) * cot).sum().backward()
        return xd.grad, m.beta.grad.clone(), m.gamma.grad.clone()
    ref = [t.clone() for t in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=d)
    a = torch.randn(4096, 4096, device=d)
    for rep in range(10):
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                a = torch.tanh(a @ a * 1e-3)
        got = run()
        assert all(torch.equal(p, q) for p, q in zip(got, ref)), f"run {rep} differs from the unloaded run"
    torch.cuda.synchronize()


@pytest.mark.parametrize("c", [192, 100])
def test_gdn_gamma_gradient_routes_agree(c, monkeypatch):
    """The persistent gamma-gradient kernel (default) against the route through crdr_conv2d_wgrad with CRDR_WGRAD_SQUARE_Q
    (CRDR_GDN_DGAMMA=wgrad): same sums in a different order."""
    from crdr_amd.models.layer.gdn import GDN
    d = dev()
    m = GDN(c)
    g = torch.Generator().manual_seed(23 + c)
    with torch.no_grad():
        m.gamma.copy_(torch.sqrt(torch.rand(c, c, generator=g) * 0.02 + 2.0 ** -36))
        m.beta.copy_(torch.sqrt(torch.rand(c, generator=g) + 0.5))
    m.to(d)
    x = (torch.randn(2, c, 90, 90, generator=g) * 2.0).to(d).contiguous(memory_format=torch.channels_last)
    cot = torch.randn(2, c, 90, 90, generator=g).to(d).contiguous(memory_format=torch.channels_last)

    def run():
        m.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        (m(xd) * cot).sum().backward()
        torch.cuda.synchronize()
        return xd.grad.clone(), m.beta.grad.clone(), m.gamma.grad.clone()
    own = run()
    monkeypatch.setenv("CRDR_GDN_DGAMMA", "wgrad")
    via = run()
    assert torch.equal(own[0], via[0])   # dx does not depend on the route
    assert rel(own[1], via[1]) < 1e-6 and rel(own[2], via[2]) < 1e-5, (rel(own[1], via[1]), rel(own[2], via[2]))
