"""Parity at BASELINE.json's full sizes (bs 16, 256x256 crops; Kodak-sized codec input) through size-independent
properties, where the CPU oracle would take minutes per case:

* linearity of the conv family in its input and in its weights at the largest stage-3 shapes (forward, input gradient,
  weight gradient through the deferred / tap-folded paths) -- a wrong tile, split or zero-fill shows up as a residual;
* encode -> .bin -> decode round trip at 768x512 reproduces y_hat / z_hat bit for bit and the same bytes decode to the
  same image;
* a full stage-3 step is bit-reproducible: two trainers built from the same seed log identical losses (fixed-order
  reductions, no float atomics), eager and graph-replayed alike.
"""
import pytest
import torch

from tests.golden.seeded_weights import seeded_input

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU test needs a HIP device"
    return torch.device("cuda:0")


FULL = [
    # name, transposed, Cin, Cout, k, stride, pad, out_pad, H (bs 16)
    ("dec 128->128 k3 @128", False, 128, 128, 3, 1, 1, 0, 128),
    ("enc 192->192 k5s2 @128", False, 192, 192, 5, 2, 2, 0, 128),
    ("dec up3 T256->256 k5s2 @64", True, 256, 256, 5, 2, 2, 1, 64),
    ("dec up4 T256->3 k5s2 @128", True, 256, 3, 5, 2, 2, 1, 128),
    ("D 3->64 k3 @256", False, 3, 64, 3, 1, 1, 0, 256),
    ("charm 480->224 k5 @16", False, 480, 224, 5, 1, 2, 0, 16),
]


@pytest.mark.parametrize("case", FULL, ids=[c[0] for c in FULL])
def test_conv_family_is_linear_at_full_size(case):
    from crdr_amd.hip import functional as HF
    name, tr, ci, co, k, s, p, op, h = case
    d = dev()
    g = torch.Generator().manual_seed(5)
    spec = HF.ConvSpec(ci, co, k, s, p, transposed=tr, out_pad=op)
    wshape = (ci, co, k, k) if tr else (co, ci, k, k)
    w1 = torch.nn.Parameter((torch.rand(wshape, generator=g) - 0.5).to(d) * (ci * k * k) ** -0.5)
    w2 = torch.nn.Parameter((torch.rand(wshape, generator=g) - 0.5).to(d) * (ci * k * k) ** -0.5)
    b0 = torch.nn.Parameter(torch.zeros(co, device=d))
    x1 = (torch.rand(16, ci, h, h, generator=g) - 0.5).to(d).requires_grad_(True)
    x2 = (torch.rand(16, ci, h, h, generator=g) - 0.5).to(d).requires_grad_(True)
    a, b = 0.75, -1.5

    def run(x, w):
        x = x.detach().requires_grad_(True)
        w = torch.nn.Parameter(w.detach().clone())
        y = HF.fused_conv(x, w, b0, spec)
        gy = torch.ones_like(y) * 0.01 + y.detach() * 0  # fixed upstream gradient
        gy = (seeded_input("gy" + name, (1, y.shape[1], 1, 1)).to(d) * 0.01).expand_as(y).contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        return y.detach(), x.grad.detach(), w.grad.detach()

    y1, dx1, dw1 = run(x1, w1)
    y2, dx2, dw2 = run(x2, w1)
    y12, dx12, dw12 = run(a * x1.detach() + b * x2.detach(), w1)
    y1w2, dx1w2, _ = run(x1, w2)
    y1w12, dx1w12, _ = run(x1, a * w1.detach() + b * w2.detach())

    def rel(u, v):
        return float((u - v).abs().max() / (v.abs().max() + 1e-20))
    assert rel(y12, a * y1 + b * y2) < 2e-5, ("forward linear in x", rel(y12, a * y1 + b * y2))
    assert rel(y1w12, a * y1 + b * y1w2) < 2e-5, ("forward linear in w", rel(y1w12, a * y1 + b * y1w2))
    assert rel(dx1w12, a * dx1 + b * dx1w2) < 2e-5, ("input gradient linear in w", rel(dx1w12, a * dx1 + b * dx1w2))
    assert rel(dx12, dx1) < 1e-6 and rel(dx2, dx1) < 1e-6, "input gradient must not depend on x"
    assert rel(dw12, a * dw1 + b * dw2) < 5e-5, ("weight gradient linear in x", rel(dw12, a * dw1 + b * dw2))


def test_codec_round_trip_at_kodak_size(tmp_path):
    from crdr_amd.utils.codec_utils import load_byte_strings, save_byte_strings
    from tests.test_gpu_model import _full_model
    model, _ = _full_model(True)
    model.eval()
    model.codec_setup()
    x = seeded_input("kodak", (1, 3, 512, 768))
    out = model.compress(x, rate_ind=1.75)
    p = tmp_path / "k.bin"
    save_byte_strings(str(p), out["string_list"])
    fake, z_hat, y_hat = model.decompress(load_byte_strings(str(p)), beta=3.84)
    assert torch.equal(y_hat.cpu(), out["y_hat"].cpu()) and torch.equal(z_hat.cpu(), out["z_hat"].cpu())
    fake2, _, _ = model.decompress(load_byte_strings(str(p)), beta=3.84)
    assert torch.equal(fake.cpu(), fake2.cpu()) and fake.shape == (1, 3, 512, 768)


def _smooth_image(h, w, seed):
    """seeded smooth noise (stand-in for a CLIC photograph: the dataset is not available offline)"""
    g = torch.Generator().manual_seed(seed)
    small = torch.rand(1, 3, h // 16 + 1, w // 16 + 1, generator=g) * 2 - 1
    return torch.nn.functional.interpolate(small, size=(h, w), mode="bicubic", align_corners=False).clamp(-1, 1)


FULLRES_WORKER = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %(root)r)
from tests.test_gpu_fullsize import _smooth_image
from tests.test_gpu_model import _full_model
model, _ = _full_model(True)
model.eval(); model.codec_setup()
out = {}
for (h, w) in ((1365, 2048), (2048, 1365)):
    x = _smooth_image(h, w, h)
    for q in (0.0, 2.25, 4.0):
        s = model.compress(x, rate_ind=q)["string_list"]
        out["%%dx%%d q%%g" %% (h, w, q)] = [hashlib.sha256(b).hexdigest() for b in s]
print("DIGESTS " + json.dumps(out))
"""


def test_fullres_codec_sweep_config5(tmp_path):
    """BASELINE config #5 stand-in (CLIC-sized inputs, both orientations, q in {0, 2.25, 4}, beta in {0, 3.84}; bpp does not
    depend on beta, so one encode serves both decodes, README.md:76): the decoder reproduces y_hat and z_hat bit for bit,
    the same bytes decode to the same image, a second PROCESS produces the same bytes, and the wall-time split
    {transforms, Charm, rANS} is reported."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    from tests.test_gpu_model import _full_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model, _ = _full_model(True)
    model.eval()
    model.codec_setup()
    digests, report = {}, []
    for (h, w) in ((1365, 2048), (2048, 1365)):
        x = _smooth_image(h, w, h)
        for q in (0.0, 2.25, 4.0):
            model.codec_profile = {}
            out = model.compress(x, rate_ind=q)
            enc = dict(model.codec_profile)
            strings = out["string_list"]
            digests["%dx%d q%g" % (h, w, q)] = [hashlib.sha256(b).hexdigest() for b in strings]
            for beta in (0.0, 3.84):
                model.codec_profile = {}
                fake, z_hat, y_hat = model.decompress(strings, beta=beta)
                dec = dict(model.codec_profile)
                assert torch.equal(y_hat, out["y_hat"]) and torch.equal(z_hat, out["z_hat"]), (h, w, q, beta)
                assert fake.shape == (1, 3, h, w) and bool(torch.isfinite(fake).all())
                report.append({"size": [h, w], "q": q, "beta": beta, "bytes": sum(len(s) for s in strings) + 12,
                               "compress_s": {k: round(v, 4) for k, v in enc.items()}, "decompress_s": {k: round(v, 4) for k, v in dec.items()}})
                assert set(enc) >= {"transforms", "charm", "rans"} and set(dec) >= {"transforms", "charm", "rans"}
            model.codec_profile = None
            fake2, _, _ = model.decompress(strings, beta=3.84)
            assert torch.equal(fake, fake2)
    print("FULLRES " + json.dumps(report))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "fullres_sweep.json"), "w") as f:
        json.dump(report, f, indent=1)
    r = subprocess.run([sys.executable, "-c", FULLRES_WORKER % {"root": root}], capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")][0]
    assert json.loads(line[8:]) == digests, "a second process produced different bytes"


@pytest.mark.parametrize("graphs", [False, True])
def test_full_stage3_step_is_bit_reproducible(graphs):
    import bench
    logs = []
    for _ in range(2):
        tr = bench.build_trainer(3, 16, 256, "cuda:0", graphs=graphs)
        tr.loss_huge_threshold = float("inf")
        loader = iter(tr.train_loader)
        out = []
        for it in range(1, 5):
            out.append(tr.optimize_parameters(it, {**next(loader), "rate_ind": 2, "beta": 0.0512 * (10 + it)}))
        logs.append(out)
        del tr
        torch.cuda.empty_cache()
    for a, b in zip(*logs):
        assert a is not None and b is not None
        assert a.keys() == b.keys()
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])
