"""CPU checks of the float64 references tests/test_gpu_tuned_plans.py replays the shipped perf database against: the sampled-pixel
convolution (both directions of crdr_conv_desc.transposed, both weight-pack layouts), the weight-gradient sample and the key
parser must agree with stock torch (F.conv2d / F.conv_transpose2d and autograd) before they may judge the HIP plans."""
import torch
import torch.nn.functional as F

from tests import test_gpu_tuned_plans as R


def _pack(w_oihw):   # [O][I][kh][kw] -> [T][O][I] (the library's forward pack, include/crdr_hip.h crdr_conv_desc.wrows / wcols)
    o, i, kh, kw = w_oihw.shape
    return w_oihw.permute(2, 3, 0, 1).reshape(kh * kw, o, i).contiguous()


def test_sampled_conv_reference_equals_torch():
    g = torch.Generator().manual_seed(0)
    for (n, c, h, w, oc, k, s, p) in [(3, 8, 11, 9, 5, 3, 1, 1), (2, 4, 12, 10, 6, 5, 2, 2), (2, 12, 7, 7, 4, 1, 1, 0)]:
        x = torch.rand((n, h, w, c + 4), generator=g) - 0.5           # pixel stride > channels
        wt = torch.rand((oc, c, k, k), generator=g) - 0.5
        ref = F.conv2d(x[..., :c].permute(0, 3, 1, 2).double(), wt.double(), stride=s, padding=p)
        oh, ow = ref.shape[2:]
        pix = R._sample_pixels(n, oh, ow)
        got = R._conv_ref64(x, _pack(wt), (n, h, w, c, oh, ow, oc, k, k, s, p, 0, 0), pix)
        assert torch.allclose(got, ref.permute(0, 2, 3, 1)[pix[0], pix[1], pix[2]], atol=1e-12)
        # transposed = 1 (ConvT forward / Conv2d input gradient): weight [I][O][kh][kw], element [t][oc][c] = w[c][oc][t]
        wt2 = torch.rand((c, oc, k, k), generator=g) - 0.5
        op = 1 if s == 2 else 0
        ref = F.conv_transpose2d(x[..., :c].permute(0, 3, 1, 2).double(), wt2.double(), stride=s, padding=p, output_padding=op)
        oh, ow = ref.shape[2:]
        pix = R._sample_pixels(n, oh, ow)
        pack = wt2.permute(2, 3, 1, 0).reshape(k * k, oc, c).contiguous()
        got = R._conv_ref64(x, pack, (n, h, w, c, oh, ow, oc, k, k, s, p, 1, 0), pix)
        assert torch.allclose(got, ref.permute(0, 2, 3, 1)[pix[0], pix[1], pix[2]], atol=1e-12)
    # tap-major pack (RGB inputs): [1][O][4 * tap + c]
    x = torch.rand((2, 9, 9, 4), generator=g) - 0.5
    x[..., 3] = 0
    wt = torch.rand((6, 3, 3, 3), generator=g) - 0.5
    pack = torch.zeros((1, 32, 64))
    pack[0, :6, :36].view(6, 9, 4)[..., :3] = wt.permute(0, 2, 3, 1).reshape(6, 9, 3)
    ref = F.conv2d(x[..., :3].permute(0, 3, 1, 2).double(), wt.double(), padding=1)
    pix = R._sample_pixels(2, 9, 9)
    got = R._conv_ref64(x, pack, (2, 9, 9, 4, 9, 9, 6, 3, 3, 1, 1, 0, 1), pix)
    assert torch.allclose(got, ref.permute(0, 2, 3, 1)[pix[0], pix[1], pix[2]], atol=1e-12)


def test_sampled_pixels_cover_the_borders_and_enough_pixels():
    for n, oh, ow in [(16, 128, 128), (16, 16, 16), (32, 256, 256), (8, 4, 4), (16, 64, 64)]:
        pn, py, px = R._sample_pixels(n, oh, ow)
        assert pn.numel() >= min(4096, n * oh * ow)
        assert {0, oh - 1} <= set(py.tolist()) and {0, ow - 1} <= set(px.tolist()) and {0, n - 1} <= set(pn.tolist())


def test_wgrad_reference_equals_autograd():
    g = torch.Generator().manual_seed(1)
    for (n, ci, h, w, co, k, s, p) in [(2, 6, 9, 8, 5, 3, 1, 1), (2, 5, 10, 10, 7, 5, 2, 2), (3, 4, 6, 6, 4, 1, 1, 0)]:
        x = (torch.rand((n, ci, h, w), generator=g) - 0.5).double()
        wt = (torch.rand((co, ci, k, k), generator=g) - 0.5).double().requires_grad_(True)
        y = F.conv2d(x, wt, stride=s, padding=p)
        dy = (torch.rand(y.shape, generator=g) - 0.5).double()
        y.backward(dy)
        P = dy.permute(0, 2, 3, 1).float().contiguous()     # dense operand = output gradient
        Q = x.permute(0, 2, 3, 1).float().contiguous()      # gathered operand = layer input
        isel, jsel = list(range(co)), list(range(ci))
        got = R._wgrad_ref64(P, Q, (n, y.shape[2], y.shape[3], h, w, k, k, s, p), isel, jsel)
        assert torch.allclose(got, wt.grad.reshape(co, ci, k * k), atol=1e-6)


def test_every_database_key_parses_into_a_known_kind():
    kinds = {}
    for kind in ("c", "g", "m", "w", "wg", "wm", "ws"):
        es = R._entries(kind)
        kinds[kind] = len(es)
        for key, algo in es:
            assert R._depth(key) > 0 and algo >= 0
    import ast
    import json
    db = json.load(open(R.DB))
    assert sum(kinds.values()) == len(db["algos"]), (kinds, len(db["algos"]))
    assert {ast.literal_eval(k)[0] for k in db["algos"]} == set(kinds)
