"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/crdr_hip.h declares; the
host entropy coder (rANS + CDF quantiser) agrees byte-for-byte with the oracle's independent pure-python
restatement, round-trips, and handles the edge cases (empty input, out-of-range "bypass" symbols, zero-frequency
repair); registry / config / optimizer-builder host logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from crdr_amd.hip import lib as L
    if not os.path.exists(L.LIB_PATH):
        L.build()
    return L.load()


def test_library_exports_every_declared_symbol(lib):
    from crdr_amd.hip import lib as L
    header = open(os.path.join(ROOT, "include", "crdr_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(crdr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    missing_binding = declared - set(L.SIGNATURES)
    assert not missing_binding, f"declared in the header but not bound in lib.py: {missing_binding}"
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"libcrdr_hip.so does not export {name}"
    assert lib.crdr_arch() == b"gfx950"
    assert lib.crdr_version() >= 100


def test_product_fails_loudly_on_cpu():
    from crdr_amd.hip import lib as L, ops
    with pytest.raises(L.CrdrHipError):
        ops.nhwc(torch.zeros(1, 4, 2, 2))


def _tables(rng, ncdf=5, maxlen=40):
    from oracle import crdr_oracle as O
    cdfs = np.zeros((ncdf, maxlen + 2), dtype=np.int32)
    sizes, offs = np.zeros(ncdf, np.int32), np.zeros(ncdf, np.int32)
    for i in range(ncdf):
        n = int(rng.integers(3, maxlen))
        pmf = rng.random(n).astype(np.float32) ** 3
        pmf /= pmf.sum()
        c = O.pmf_to_quantized_cdf(list(pmf) + [1e-4], 16)
        cdfs[i, :len(c)] = c
        sizes[i] = len(c)
        offs[i] = -(n // 2)
    return cdfs, sizes, offs


def test_pmf_to_quantized_cdf_matches_oracle(lib):
    from oracle import crdr_oracle as O
    from crdr_amd.codec.tables import pmf_to_quantized_cdf
    rng = np.random.default_rng(0)
    for trial in range(40):
        n = int(rng.integers(2, 300))
        pmf = rng.random(n).astype(np.float32) ** rng.integers(1, 12)  # many (near-)zero entries -> repair loop
        if trial % 5 == 0:
            pmf[rng.integers(0, n, size=n // 2)] = 0.0
        pmf[rng.integers(0, n)] += 1.0
        pmf = (pmf / pmf.sum()).astype(np.float32)
        got = pmf_to_quantized_cdf(pmf, 16)
        ref = O.pmf_to_quantized_cdf(list(pmf), 16)
        assert list(got) == ref
        assert got[0] == 0 and got[-1] == 1 << 16 and np.all(np.diff(got) > 0)
    with pytest.raises(Exception):
        pmf_to_quantized_cdf(np.array([0.5, float("nan")], np.float32))


def test_rans_roundtrip_and_oracle_bytes(lib):
    from oracle import crdr_oracle as O
    from crdr_amd.codec import rans
    rng = np.random.default_rng(1)
    cdfs, sizes, offs = _tables(rng)
    for n in (0, 1, 7, 1000):
        idx = rng.integers(0, len(sizes), size=n).astype(np.int32)
        # symbols mostly inside the table, some far outside on both sides (bypass coding, multi-nibble escapes)
        sym = np.array([int(rng.integers(offs[i] - 2, offs[i] + sizes[i])) for i in idx], dtype=np.int32)
        if n >= 7:
            sym[::5] += rng.integers(-70000, 70000, size=len(sym[::5])).astype(np.int32)
        data = rans.encode_with_indexes(sym, idx, cdfs, sizes, offs)
        assert len(data) % 4 == 0 and len(data) >= 8
        ref = O.rans_encode(sym.tolist(), idx.tolist(), cdfs, sizes, offs)
        assert data == ref, "C coder and oracle restatement disagree"
        out = rans.decode_with_indexes(data, idx, cdfs, sizes, offs)
        assert np.array_equal(out, sym)
        assert O.RansDecoder(data).decode(idx.tolist(), cdfs, sizes, offs) == sym.tolist()
        # streaming decode in 3 chunks == one-shot (the Charm decoder reads one slice at a time)
        dec = rans.RansDecoder()
        dec.set_stream(data)
        parts = [dec.decode_stream(p, cdfs, sizes, offs) for p in np.array_split(idx, 3)]
        assert np.array_equal(np.concatenate(parts), sym)
    with pytest.raises(Exception):
        rans.decode_with_indexes(b"\x00\x01\x02", np.zeros(1, np.int32), cdfs, sizes, offs)


def test_gaussian_tables_match_oracle(lib):
    from oracle import crdr_oracle as O
    from crdr_amd.models.subnet.entropy_model.gaussian_conditional import GaussianMeanScaleConditional, get_scale_table
    m = GaussianMeanScaleConditional(scale_bound=0.11)
    m.update_scale_table(get_scale_table(), force=True)
    table, length, offset = O.gaussian_cdf_tables()
    assert np.array_equal(m._quantized_cdf.numpy(), table)
    assert np.array_equal(m._cdf_length.numpy(), length) and np.array_equal(m._offset.numpy(), offset)
    assert m._quantized_cdf.shape[0] == 64
    s = torch.tensor([[0.01, 0.11, 0.12, 1.0, 255.0, 300.0]])
    assert torch.equal(m.build_indexes(s), O.build_indexes(s))
    # known answers: the smallest scale is the bound, index 0; anything above the table's top maps to the last entry
    assert m.build_indexes(torch.tensor([0.05])).item() == 0 and m.build_indexes(torch.tensor([1e4])).item() == 63


def test_entropy_bottleneck_tables_and_schema(lib):
    from oracle import crdr_oracle as O
    from crdr_amd.models.subnet.entropy_model.entropy_bottleneck import SteEntropyBottleneck
    from tests.golden.seeded_weights import seeded_tensor
    m = SteEntropyBottleneck(channels=12)
    assert [k for k, _ in m.named_parameters()] == O.eb_param_names("")[:0] + [n[1:] for n in O.eb_param_names("")]
    sd = m.state_dict()
    ref = {}
    for k, v in sd.items():
        if torch.is_floating_point(v) and v.numel() > 0 and k != "target":
            sd[k] = seeded_tensor("entropy_model_z." + k, v.shape)
            ref["entropy_model_z." + k] = sd[k]
    m.load_state_dict(sd)
    assert m.update(force=True)
    table, length, offset = O.eb_cdf_tables(ref)
    assert np.array_equal(m._quantized_cdf.numpy(), table)
    assert np.array_equal(m._cdf_length.numpy(), length) and np.array_equal(m._offset.numpy(), offset)
    # compress / decompress of integers around the medians round-trips through the C coder
    z = torch.round(torch.randn(2, 12, 3, 5) * 4) + m._get_medians().detach().reshape(1, -1, 1, 1)
    strings = m.compress(z)
    sym = m.decompress(strings, (3, 5))
    assert torch.allclose(m.dequantize(sym), z)
    assert m.packed_params().shape == (12, 58)


def test_registry_and_builders():
    from crdr_amd.utils.registry import Registry
    r = Registry("x")

    @r.register()
    class Foo:
        pass
    assert r.get("Foo") is Foo and "Foo" in r and r.keys() == ["Foo"]
    with pytest.raises(AssertionError):
        r.register()(Foo)
    with pytest.raises(KeyError):
        r.get("Bar")
    import crdr_amd.losses, crdr_amd.models, crdr_amd.trainer  # noqa: F401
    from crdr_amd.utils import registry as R
    assert {"HyperpriorCharmModel", "InterpCaHyperpriorCharmModel", "BetaCondInterpCaHyperpriorCharmModel"} <= set(R.MODEL_REGISTRY.keys())
    assert {"ElicEncoder", "ElicInterpCaEncoder"} <= set(R.ENCODER_REGISTRY.keys())
    assert {"ElicDecoder", "ElicInterpCaDecoder", "ElicInterpCaBetaCondDecoder"} <= set(R.DECODER_REGISTRY.keys())
    assert {"RateDistortionTrainer", "MultirateBetaCondHrrGanRateDistortionTrainer"} <= set(R.TRAINER_REGISTRY.keys())
    assert {"MSELoss", "HificRateLoss", "HificVariableRateLoss", "LPIPSLoss", "VanillaGANLoss"} <= set(R.LOSS_REGISTRY.keys())
    assert {"ModuleListDiscriminator", "CLIC21GVAEDiscriminator"} <= set(R.DISCRIMINATOR_REGISTRY.keys())
    assert "Adam" in R.OPTIMIZER_REGISTRY and "MultiStepLR" in R.SCHEDULER_REGISTRY


def test_train_config_cli_overlay(tmp_path, monkeypatch):
    from crdr_amd.utils.options import ConfigDict, TrainConfig
    monkeypatch.chdir(ROOT)
    os.makedirs(tmp_path / "ck", exist_ok=True)
    cfgdir = tmp_path / "cfg"
    os.makedirs(cfgdir)
    (cfgdir / "base.yaml").write_text("ckpt_root: %s\na: {x: 1, y: 2}\ndataset: {batch_size: 8}\n" % (tmp_path / "ck"))
    (cfgdir / "exp7.yaml").write_text("_base_: ./base.yaml\na: {y: 3}\nb: {_delete_: true, k: 1}\n")
    opt = TrainConfig.get_opt(argv=[str(cfgdir / "exp7.yaml"), "-b", "16", "-d", "cuda:3", "-ti", "100"])
    assert opt.exp == "exp7" and opt.device == "cuda:3" and opt.total_iter == 100
    assert opt.dataset.batch_size == 16 and opt.a.x == 1 and opt.a.y == 3 and opt.is_train
    assert opt.path.model_dir.endswith(os.path.join("exp7", "model"))
    with pytest.raises(AttributeError):
        opt.nope
    assert opt.get("nope", 5) == 5
    c = ConfigDict({"p": {"q": [1, {"r": 2}]}})
    assert c.p.q[1].r == 2 and c.to_dict() == {"p": {"q": [1, {"r": 2}]}}
    opt.dump(str(tmp_path / "dump.yaml"))
    assert "total_iter: 100" in open(tmp_path / "dump.yaml").read()


def test_multistep_lr_and_rate_loss_host_logic():
    from crdr_amd.losses.rate_loss import HificRateLoss, HificVariableRateLoss
    from crdr_amd.trainer.optimizer.build_optimizer_scheduler import MultiStepLR

    class FakeOpt:
        param_groups = [{"lr": 1e-4}]
    s = MultiStepLR(FakeOpt(), milestones=[3], gamma=0.1)
    lrs = []
    for _ in range(4):
        s.step()
        lrs.append(FakeOpt.param_groups[0]["lr"])
    assert lrs[:2] == [1e-4, 1e-4] and abs(lrs[2] - 1e-5) < 1e-12 and abs(lrs[3] - 1e-5) < 1e-12
    bpp, qbpp = torch.tensor([0.31, 0.52]), torch.tensor([0.29, 0.49])
    vr = HificVariableRateLoss(lambda_A=[3.6, 1.8, 0.8, 0.4, 0.1], lambda_B=0.015625, target_rate=[0.08, 0.16, 0.36, 0.72, 1.2])
    assert abs(vr(bpp, qbpp=qbpp, current_iter=1, rate_ind=torch.tensor([2])).item() - 0.8 * 0.415) < 1e-6
    assert abs(vr(bpp, qbpp=qbpp, current_iter=1, rate_ind=3).item() - 0.015625 * 0.415) < 1e-7
    assert abs(HificRateLoss(0.05, 0.015625, 1.5)(bpp, qbpp=qbpp, current_iter=1).item() - 0.015625 * 0.415) < 1e-7
    with pytest.raises(AssertionError):
        HificVariableRateLoss(lambda_A=[1.0, 2.0], lambda_B=0.1, target_rate=[0.1, 0.2])


def test_calc_metrics_psnr_semantics(tmp_path):
    """scripts/calc_metrics.py PSNR: uint8 read-back, float32 squared error, per-image PSNR then mean
    (reference scripts/calc_metrics.py:146,168) -- not the PSNR of the pooled MSE."""
    import importlib.util
    import numpy as np
    from PIL import Image
    spec = importlib.util.spec_from_file_location("calc_metrics", os.path.join(ROOT, "scripts", "calc_metrics.py"))
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    rng = np.random.default_rng(0)
    rd, fd = tmp_path / "real", tmp_path / "fake"
    rd.mkdir(), fd.mkdir()
    expect = []
    for i, noise in enumerate((2, 20)):
        a = rng.integers(0, 256, size=(16, 24, 3), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-noise, noise + 1, size=a.shape), 0, 255).astype(np.uint8)
        Image.fromarray(a).save(rd / f"im{i}.png")
        Image.fromarray(b).save(fd / f"im{i}.png")
        mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
        expect.append(10 * np.log10(255.0 ** 2 / mse))
    res = cm.main(["--real_dir", str(rd), "--fake_dir", str(fd), "--metrics", "psnr"])
    assert res["num_images"] == 2 and abs(res["PSNR"] - float(np.mean(expect))) < 1e-3
    pooled = 10 * np.log10(255.0 ** 2 / np.mean([255.0 ** 2 / 10 ** (e / 10) for e in expect]))
    assert abs(res["PSNR"] - pooled) > 1.0  # averaging per-image PSNRs is a different number


def test_multistep_lr_is_chainable_like_torch():
    """an lr set from outside (load_checkpoint new_g_lr) must survive scheduler.step() until the next
    milestone, exactly like torch.optim.lr_scheduler.MultiStepLR (the class the reference builds)."""
    import torch
    from crdr_amd.trainer.optimizer.build_optimizer_scheduler import MultiStepLR

    class _Opt:
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]
    ours = _Opt(1e-4)
    sch = MultiStepLR(ours, milestones=[3, 6], gamma=0.1)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.Adam([p], lr=1e-4)
    ref = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones=[3, 6], gamma=0.1)
    for it in range(1, 9):
        if it == 2:  # resume-style override
            ours.param_groups[0]["lr"] = 5e-5
            ref_opt.param_groups[0]["lr"] = 5e-5
        ref_opt.step()
        sch.step()
        ref.step()
        assert abs(ours.param_groups[0]["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12, it
    # state round trip keeps counters, not the lr
    sd = sch.state_dict()
    o2 = _Opt(7e-6)
    s2 = MultiStepLR(o2, milestones=[1], gamma=0.5)
    s2.load_state_dict(sd)
    assert s2.last_epoch == 8 and s2.milestones == [3, 6] and o2.param_groups[0]["lr"] == 7e-6


def test_lpips_loss_refuses_random_weights_unless_allowed(tmp_path, monkeypatch):
    import torch
    from crdr_amd.losses.perceptual_loss import ALEX_CFG, LPIPSLoss, _TV_IDX
    monkeypatch.delenv("CRDR_ALLOW_RANDOM_LPIPS", raising=False)
    monkeypatch.delenv("CRDR_LPIPS_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="pretrained"):
        LPIPSLoss(loss_weight=1.0)
    assert LPIPSLoss(loss_weight=1.0, allow_random_weights=True).pretrained is False
    feats, lin = {}, {}
    for i, (ci, co, k, _, _) in enumerate(ALEX_CFG):
        feats[f"{_TV_IDX[i]}.weight"] = torch.full((co, ci, k, k), 0.01 * (i + 1))
        feats[f"{_TV_IDX[i]}.bias"] = torch.full((co,), 0.1 * (i + 1))
        lin[f"lin{i}.model.1.weight"] = torch.full((1, co, 1, 1), float(i + 2))
    f = tmp_path / "lpips_alex.pth"
    torch.save({"alexnet_features": feats, "lpips_lin": lin}, f)
    m = LPIPSLoss(loss_weight=1.0, weights=str(f))
    assert m.pretrained and float(m.lpips.net[2].bias[0]) == pytest.approx(0.3) and float(m.lpips.lin[4][7]) == 6.0
    monkeypatch.setenv("CRDR_LPIPS_WEIGHTS", str(f))
    assert LPIPSLoss(loss_weight=1.0).pretrained
    # scripts/calc_metrics.py reads the same file layout
    from crdr_amd.losses.perceptual_loss import LpipsAlex
    n = LpipsAlex()
    n.load_lpips_file(str(f))
    assert float(n.net[0].weight[0, 0, 0, 0]) == pytest.approx(0.01)
    bad = tmp_path / "bad.pth"
    torch.save({"x": 1}, bad)
    with pytest.raises(ValueError):
        n.load_lpips_file(str(bad))


def test_eval_grid_csv_layout_and_gate(tmp_path):
    """scripts/eval_grid.py writes the reference's rd_results column layout and applies BASELINE.json's gate
    (bpp +-1e-4, PSNR +-0.01 dB) row by row."""
    import csv
    from scripts import eval_grid as E
    assert E.COLUMNS == ["dataset", "quality", "beta", "bpp", "PSNR", "LPIPS", "DISTS"]
    assert E.default_qualities() == [0.25 * i for i in range(17)]
    rows = [{"dataset": "kodak", "quality": 0.0, "beta": 3.84, "bpp": 0.10945, "PSNR": 27.4451, "LPIPS": "", "DISTS": ""},
            {"dataset": "kodak", "quality": 4.0, "beta": 0.0, "bpp": 1.0501, "PSNR": 37.70, "LPIPS": 0.1, "DISTS": ""}]
    out = tmp_path / "k.csv"
    E.write_csv(rows, str(out))
    with open(out) as f:
        got = list(csv.reader(f))
    assert got[0] == E.COLUMNS and got[1][:3] == ["kodak", "0.0", "3.84"] and len(got) == 3
    ref = tmp_path / "ref.csv"
    with open(ref, "w") as f:
        f.write("dataset,quality,beta,bpp,PSNR,LPIPS,DISTS\nkodak,0.0,3.84,0.10944959852430554,27.44512440303645,0.0963,0.104\n"
                "kodak,4.0,0.0,1.0503268771701388,37.640484422314252,0.03,0.04\nkodak,2.0,0.0,0.4,32.9,0.1,0.1\n")
    cmp = E.compare_with_reference(rows, str(ref))
    assert [(c[0], c[1], c[4]) for c in cmp] == [(0.0, 3.84, True), (4.0, 0.0, False)]


def test_shipped_perf_database_matches_the_library():
    """crdr_amd/hip/tune_gfx950.json is keyed by the library version and its configuration counts: a database of another build is
    silently ignored at start-up (every shape would be tuned live), so a stale one must not be committed; its algorithm ids must
    also be ids this library knows."""
    import json
    from crdr_amd.hip import lib as L
    from crdr_amd.hip import ops
    db = json.load(open(ops.DEFAULT_TUNE_DB))
    assert db["signature"] == ops._tune_signature(), (db["signature"], ops._tune_signature())
    lib = L.load()
    nconv = lib.crdr_conv2d_num_configs() + lib.crdr_conv2d_num_stream_configs() + lib.crdr_conv2d_num_wino_configs()
    nw = lib.crdr_conv2d_wgrad_num_configs() + lib.crdr_conv2d_wgrad_num_wino_configs() - 1   # (the F(3x3, 4x4) slab kernel: the id behind the last configuration)
    assert len(db["algos"]) > 300
    for k, v in db["algos"].items():
        kind = k[2:k.index("'", 2)]
        cfg = v & 0xff
        assert 0 <= cfg <= (nw if kind in ("w", "wg", "ws", "wm") else nconv), (k, v)


def test_committed_traffic_measurement_is_of_this_library():
    """bench.py prices roofline.traffic with the newest profiles/r*_hbm_families.json and refuses one measured with another
    crdr_version(): the committed one must be of the committed library."""
    import glob
    import json
    from crdr_amd.hip import lib as L
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_families.json")))
    assert fs, "no PMC family file committed"
    doc = json.load(open(fs[-1]))
    assert doc.get("library_version") == int(L.load().crdr_version()), (fs[-1], doc.get("library_version"))
    assert "conv_fwd_dgrad" in doc["families"]


def test_winograd_ids_are_planned_only_for_the_shapes_they_take(lib):
    """Host-side planning of the Winograd variants (no GPU needed): the forced ids are accepted for 3x3 / 5x5 stride-1 convolutions
    and their stride-1 transposed twins, rejected otherwise; the pair-tile variant needs a channel tail of 1..32 and more than one
    16x16 patch; the workspace holds the tickets + 16 positions x (sub-filters) x padded OC x padded C floats; the last
    weight-gradient configuration is planned for 3x3 stride 1 only."""
    import ctypes as C
    from crdr_amd.hip import lib as L
    base = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs()
    assert lib.crdr_conv2d_num_wino_configs() == 3

    def desc(c, oc, h, k, stride=1, pad=None, transposed=0, n=2):
        pad = k // 2 if pad is None else pad
        oh = (h - 1) * stride - 2 * pad + k if transposed else (h + 2 * pad - k) // stride + 1
        cols = (c + 31) // 32 * 32
        return L.ConvDesc(N=n, H=h, W=h, C=c, OH=oh, OW=oh, OC=oc, kh=k, kw=k, stride=stride, pad=pad, transposed=transposed, ldx=c, ldy=oc,
                          wrows=oc, wcols=cols, flags=0, ldres=0, ldg=0, wlayout=0, reserved=0, ldpre=0, ldmask=0)

    def plan(d, algo, g=1):
        d.reserved = algo
        return lib.crdr_conv2d_choose_algo(C.byref(d), g)

    assert plan(desc(96, 96, 32, 3), base) == base
    assert plan(desc(96, 96, 32, 3), base + 1) == base + 1            # tail 32, 2 x 4 patches
    assert plan(desc(128, 128, 32, 3), base + 1) == 0                 # no channel tail
    assert plan(desc(96, 104, 32, 3), base + 1) == 0                  # tail 40 > 32
    assert plan(desc(96, 96, 16, 3, n=1), base + 1) == 0              # a single patch: nothing to pair
    assert plan(desc(96, 96, 32, 3, transposed=1), base) == base      # stride-1 transposed = input gradient
    assert plan(desc(320, 224, 16, 5), base) == base                  # 5x5 as 2 x 2 sub-filters
    assert plan(desc(96, 96, 32, 3, stride=2), base) == 0
    assert plan(desc(96, 96, 32, 1), base) == 0
    assert plan(desc(96, 96, 32, 7), base) == 0
    d = desc(100, 96, 32, 3)
    d.reserved = base
    tickets = 16384 * 4
    assert lib.crdr_conv2d_workspace(C.byref(d)) == tickets + 2 * 13 * 16 * 2 * 64 * 16   # 2 N tiles x 13 chunks of 8 channels x 2048 slots of 16 B
    d5 = desc(32, 64, 16, 5)
    d5.reserved = base
    assert lib.crdr_conv2d_workspace(C.byref(d5)) == tickets + 1 * 4 * 4 * 16 * 2 * 64 * 16
    # variant 2 = F(4x4, 3x3) (wino4.hip): 3x3 stride 1 with >= 24 output columns (16 x 32-pixel tiles up to 32 columns, 8 x 64 beyond), C and
    # OC multiples of 4; workspace =
    # tickets + N tiles of 64 x chunks of 4 channels x 36 positions x 4 x 64 floats
    assert plan(desc(96, 96, 64, 3), base + 2) == base + 2
    assert plan(desc(96, 96, 64, 3, transposed=1), base + 2) == base + 2
    assert plan(desc(96, 96, 32, 3), base + 2) == base + 2            # 32 output columns: the 16 x 32 tile geometry
    assert plan(desc(96, 96, 16, 3), base + 2) == base + 2            # 16 x 16 images: two whole images per tile
    assert plan(desc(96, 96, 8, 3), base + 2) == 0                    # 8 x 8 images
    assert plan(desc(96, 96, 64, 3, stride=2), base + 2) == 0
    assert plan(desc(320, 224, 64, 5), base + 2) == base + 2          # 5x5 stride 1 pad 2: four shifted 3x3 sub-filters
    assert plan(desc(320, 224, 16, 5), base + 2) == base + 2          # (the 16 x 16 stage of the context model)
    assert plan(desc(8, 224, 64, 5), base + 2) == 0                   # ... of >= 12 input channels
    # 5x5 stride 2 pad 2: conv as four parity sub-filters (even input, >= 24 output columns), transposed conv as four output phases
    assert plan(desc(192, 192, 128, 5, stride=2), base + 2) == base + 2
    assert plan(desc(192, 192, 64, 5, stride=2), base + 2) == base + 2  # 32 output columns
    assert plan(desc(192, 192, 32, 5, stride=2), base + 2) == 0       # 16 output columns
    dt = desc(256, 256, 64, 5, stride=2, transposed=1)
    dt.OH = dt.OW = 128                                               # output_padding 1
    assert plan(dt, base + 2) == base + 2
    dt.reserved = base + 2
    assert lib.crdr_conv2d_workspace(C.byref(dt)) == 16384 * 4 + 4 * 4 * 64 * 36 * 4 * 64 * 4   # 4 N tiles x 4 phases x 64 chunks x 36 KiB
    assert plan(desc(96, 98, 64, 3), base + 2) == 0                   # OC % 4 != 0
    # K splits of the F(4x4) kernel: bits 8..11 of the id = splits - 1; workspace + tiles x splits x 128 KiB of partial tiles
    hoist_t = desc(4256, 320, 16, 5, transposed=1, n=16)              # the hoisted convs' input gradient: 8 image pairs x 5 N tiles = 40 tiles
    assert plan(hoist_t, (base + 2) | (5 << 8)) == (base + 2) | (5 << 8)
    u_bytes = 5 * 4 * 1064 * 36 * 4 * 64 * 4                          # 5 N tiles x 4 sub-filters x 1 064 chunks x 36 KiB
    hoist_t.reserved = base + 2
    assert lib.crdr_conv2d_workspace(C.byref(hoist_t)) == tickets + u_bytes
    hoist_t.reserved = (base + 2) | (5 << 8)
    assert lib.crdr_conv2d_workspace(C.byref(hoist_t)) == tickets + u_bytes + 40 * 6 * 131072
    assert plan(desc(12, 64, 64, 3), (base + 2) | (3 << 8)) == 0      # 3 sub-steps cannot feed 4 splits
    assert plan(desc(12, 64, 64, 3), (base + 2) | (2 << 8)) == (base + 2) | (2 << 8)
    assert plan(desc(24, 64, 16, 5), (base + 2) | (6 << 8)) == 0      # 24 sub-steps in 7 parts of 4: the last one would be empty
    d4 = desc(100, 96, 64, 3)
    d4.reserved = base + 2
    assert lib.crdr_conv2d_workspace(C.byref(d4)) == tickets + 2 * 25 * 36 * 4 * 64 * 4
    nw = lib.crdr_conv2d_wgrad_num_configs()

    def wdesc(k, stride):
        return L.WgradDesc(N=2, PH=32, PW=32, PC=96, ldp=96, QH=32, QW=32, QC=64, ldq=64, kh=k, kw=k, stride=stride, pad=k // 2, gI=96, gJ=64, accumulate=0, algo=nw)
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(wdesc(3, 1))) == 9 * 96 * 64 * 4      # unsplit: one slab of 9 taps
    # the F(3x3, 4x4) slab kernel: the id behind the last configuration; 3x3 stride 1 and 5x5 pad 2 (stride 1 / 2) with QC > 4
    assert lib.crdr_conv2d_wgrad_num_wino_configs() == 2

    def w4desc(k, stride, algo):
        d = wdesc(k, stride)
        d.algo = algo
        return d
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(3, 1, nw + 1))) == 9 * 96 * 64 * 4
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(3, 1, (nw + 1) | (2 << 8)))) == 4 * 9 * 96 * 64 * 4   # 4 strip splits: 4 slabs
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(5, 1, nw + 1))) == 25 * 96 * 64 * 4
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(5, 2, nw + 1))) == 25 * 96 * 64 * 4
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(3, 2, nw + 1))) == 0                                   # 3x3 stride 2: not taken
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(w4desc(3, 1, (nw + 1) | (8 << 8)))) == 0                       # 256 splits of 2 x 8 x 2 strips
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(wdesc(5, 1))) == 0                     # rejected (0 = planning failed)
    assert lib.crdr_conv2d_wgrad_workspace(C.byref(wdesc(3, 2))) == 0


def test_filter_cache_bookkeeping():
    """ops' persistent filter caches (host side only, no launches): only registered persistent packs count as cacheable, a registration dies
    with its tensor, every writer of a pack bumps its version, filter_scope is a no-op kept for older callers, and dropping the caches of a
    pack drops exactly those"""
    import torch
    from crdr_amd.hip import ops
    a, b = torch.zeros(64), torch.zeros(64)
    assert not ops._is_persistent_pack(a.data_ptr())
    ops.register_persistent_pack(a)
    assert ops._is_persistent_pack(a.data_ptr()) and not ops._is_persistent_pack(b.data_ptr())
    v0 = ops.pack_version(a.data_ptr())
    ops.bump_pack_version(a.data_ptr())
    assert ops.pack_version(a.data_ptr()) == v0 + 1 and ops.pack_version(b.data_ptr()) == 0
    keep = dict(ops._filter_cache)
    try:
        ops._filter_cache.clear()
        for name, wk in (("Ua", (a.data_ptr(),)), ("Uab", (a.data_ptr(), b.data_ptr())), ("Ub", (b.data_ptr(),))):
            e = ops._FilterCache()
            e.wkeys, e.u, e.versions, e.item, e.nbytes = wk, name, None, None, 0
            ops._filter_cache[(wk, len(wk), 37)] = e
        with ops.filter_scope():
            with ops.filter_scope():
                assert len(ops._filter_cache) == 3
        serial = ops._filter_serial[0]
        ops.filter_scope_invalidate(a.data_ptr())
        assert [e.u for e in ops._filter_cache.values()] == ["Ub"] and ops._filter_serial[0] > serial
        ops.filter_scope_invalidate()
        assert ops._filter_cache == {}
    finally:
        ops._filter_cache.clear()
        ops._filter_cache.update(keep)
    ptr = a.data_ptr()
    del a
    assert not ops._is_persistent_pack(ptr)



def test_merge_tune_db_keeps_the_winograd_class(tmp_path):
    """tools/merge_tune_db.py: a candidate's entry is taken when it stays among the direct / streaming ids, kept when it would move a layer to or
    from a Winograd id (forward ids behind the streaming variants, the two Winograd weight-gradient ids), split bits ignored"""
    import json
    import subprocess
    import sys
    sig = "v500-c27-s8-w24+1-n3"   # forward Winograd ids 36..38, weight-gradient ids 24, 25
    shipped = {"signature": sig, "algos": {"('c', 1)": 5, "('c', 2)": 38, "('m', 3)": 7, "('g', 4)": 36 | (1 << 8), "('w', 5)": 3, "('wm', 6)": 25, "('ws', 7)": 9}}
    cand = {"signature": sig, "algos": {"('c', 1)": 12 | (2 << 8), "('c', 2)": 4, "('m', 3)": 38, "('g', 4)": 36 | (2 << 8), "('w', 5)": 24, "('wm', 6)": 2, "('ws', 7)": 11}}
    a, b, o = (str(tmp_path / n) for n in ("a.json", "b.json", "o.json"))
    json.dump(shipped, open(a, "w"))
    json.dump(cand, open(b, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "merge_tune_db.py"), a, b, o], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    out = json.load(open(o))["algos"]
    assert out == {"('c', 1)": 12 | (2 << 8), "('c', 2)": 38, "('m', 3)": 7, "('g', 4)": 36 | (1 << 8), "('w', 5)": 3, "('wm', 6)": 25, "('ws', 7)": 11}, out
    assert "2 entries re-timed, 5 kept" in r.stdout, r.stdout
