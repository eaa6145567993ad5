"""Narrowing the UNPINNED part of the parity claim: the entropy arithmetic lives in
compressai==1.2.4 (pyproject.toml:16), which is neither installed nor vendored here, so the oracle restates it.  What can
be checked offline is checked here, on CPU:

 (i)   the oracle's Gaussian-conditional likelihood / bits against an independent float64 evaluation (scipy.stats.norm)
       over a (sigma, |y - mu|) grid that crosses the 0.11 scale bound and the 1e-9 likelihood floor;
 (ii)  `pmf_to_quantized_cdf` -- the oracle's and the shipped C one -- against an independent exact-rational
       implementation, on all 64 Gaussian rows, the seeded factorised-prior rows and adversarial pmfs; the 64 Gaussian rows
       also against tables built from float64 probabilities (what a higher-precision compressai would produce);
 (iii) against compressai itself whenever it is importable (skipped here): likelihoods, both `update()` tables, the
       quantiser and the rANS bytes;
 (v)   `load_learned_weight` on a synthetic checkpoint with compressai-style keys / buffer shapes
       (base_model.py:80-118)."""
import math
import os
from fractions import Fraction

import numpy as np
import pytest
import torch

from oracle import crdr_oracle as O
from tests.golden.seeded_weights import seeded_tensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------- (i)
def test_gaussian_likelihood_against_float64_scipy():
    from scipy.stats import norm
    sig = torch.tensor(np.concatenate([[-1.0, 0.0, 0.01, 0.05, 0.1099, 0.11, 0.1101], np.geomspace(0.12, 256.0, 40)]), dtype=torch.float32)
    v = torch.tensor(np.concatenate([[0.0, 1e-4, 0.25, 0.4999, 0.5, 0.5001, 1.0, 1.5], np.linspace(2.0, 60.0, 59), [200.0, 1e4]]), dtype=torch.float32)
    S, Vv = torch.meshgrid(sig, v, indexing="ij")
    mu = torch.full_like(S, 0.375)
    lik = O.gaussian_likelihood(mu + Vv, mu, S).double().numpy()
    s64 = np.maximum(S.double().numpy(), 0.11)
    a = np.abs((mu + Vv).double().numpy() - mu.double().numpy())  # what fp32 inputs actually hold
    ref = norm.cdf((0.5 - a) / s64) - norm.cdf((-0.5 - a) / s64)
    ref = np.maximum(ref, 1e-9)
    big = ref > 1e-5
    assert np.all(np.abs(lik[big] - ref[big]) <= 3e-6 * ref[big] + 2e-7), float(np.max(np.abs(lik[big] - ref[big]) / ref[big]))
    # towards the floor fp32 erfc differences are absolute, not relative; the floor itself is exact
    assert np.all(np.abs(lik[~big] - ref[~big]) <= 2e-7)
    floor = float(np.float32(1e-9))  # the bound is an fp32 constant (9.99999972e-10)
    assert np.all(lik >= floor) and np.any(lik == floor), "the grid must reach the likelihood floor"
    bits, bits_ref = -np.log2(lik), -np.log2(ref)
    assert np.all(np.abs(bits[big] - bits_ref[big]) <= 1e-5 * np.maximum(1.0, bits_ref[big]) + 1e-5)
    # known answers: P(|x| < 1/2), sigma = 1  and the clamp of sigma below 0.11
    one = torch.zeros(1)
    assert abs(float(O.gaussian_likelihood(one, one, one + 1.0)) - math.erf(0.5 / math.sqrt(2))) < 1e-7
    assert torch.equal(O.gaussian_likelihood(one, one, one + 0.01), O.gaussian_likelihood(one, one, one + 0.11))


def test_lower_bound_gradient_rule():
    """compressai LowerBound: gradient passes where x >= bound OR the gradient would push x up (g < 0)."""
    x = torch.tensor([0.05, 0.05, 0.2, 0.11], requires_grad=True)
    y = O.lower_bound(x, 0.11)
    y.backward(torch.tensor([1.0, -1.0, 1.0, 1.0]))
    assert x.grad.tolist() == [0.0, -1.0, 1.0, 1.0] and y.tolist() == pytest.approx([0.11, 0.11, 0.2, 0.11])


# ---------------------------------------------------------------------------------------------------------- (ii)
def exact_pmf_to_quantized_cdf(pmf, precision=16):
    """Independent implementation in exact rational arithmetic (no float after the fp32 inputs are read):
    f_i = round_half_away(float32(p_i) * 2^precision computed in float32), rescale by integer division, prefix-sum, pin the
    end, then give every empty symbol one count taken from the least frequent symbol that can spare one."""
    scale = np.float32(1 << precision)
    freqs = []
    for p in pmf:
        prod = Fraction(float(np.float32(np.float32(p) * scale)))  # the float32 product, read exactly
        fl = prod.numerator // prod.denominator
        freqs.append(fl + (1 if prod - fl >= Fraction(1, 2) else 0))
    freqs = [0] + freqs
    total = sum(freqs)
    assert total > 0
    scaled = [((1 << precision) * f) // total for f in freqs]
    cdf, run = [], 0
    for f in scaled:
        run += f
        cdf.append(run)
    cdf[-1] = 1 << precision
    n = len(cdf)
    for i in range(n - 1):
        if cdf[i] == cdf[i + 1]:
            cands = [(cdf[j + 1] - cdf[j], j) for j in range(n - 1) if cdf[j + 1] - cdf[j] > 1]
            assert cands
            best = min(cands, key=lambda t: (t[0], t[1]))[1]  # smallest frequency, first such symbol
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return cdf


def _c_quantizer():
    from crdr_amd.codec.tables import pmf_to_quantized_cdf
    return lambda p: [int(v) for v in pmf_to_quantized_cdf(np.asarray(p, dtype=np.float32), 16)]


def _gaussian_rows():
    from scipy.stats import norm
    st = O.get_scale_table()
    mult = -norm.ppf(1e-9 / 2)
    center = torch.ceil(st * mult).int()
    rows = []
    for k in range(len(st)):
        c = int(center[k])
        s = torch.abs(torch.arange(2 * c + 1).int() - c).float()
        up, lo = O._phi((0.5 - s) / st[k]), O._phi((-0.5 - s) / st[k])
        rows.append(((up - lo).numpy().tolist() + [float(2 * lo[0])], c))
    return st, rows


def test_quantised_cdf_all_gaussian_rows_three_implementations_agree():
    cq = _c_quantizer()
    st, rows = _gaussian_rows()
    table, lengths, offsets = O.gaussian_cdf_tables()
    assert table.shape[0] == 64 == len(rows)
    for k, (pmf, c) in enumerate(rows):
        ex = exact_pmf_to_quantized_cdf(pmf)
        assert ex == O.pmf_to_quantized_cdf(pmf) == cq(pmf), f"scale level {k}"
        assert ex[0] == 0 and ex[-1] == 65536 and all(b > a for a, b in zip(ex, ex[1:])), f"row {k} not strictly increasing"
        assert int(lengths[k]) == len(ex) == 2 * c + 3 and int(offsets[k]) == -c
        assert table[k, : len(ex)].tolist() == ex and not table[k, len(ex):].any()


def test_quantised_cdf_gaussian_rows_vs_float64_probabilities():
    """Tables from float64 probabilities (scipy) differ from the fp32-built ones by at most 2 counts out of 65536, at
    < 5 % of the entries (measured: worst 2, 796 of 27 256): the fp32 pmf is not where bits are lost."""
    from scipy.stats import norm
    st, rows = _gaussian_rows()
    worst, differing, total = 0, 0, 0
    for k, (pmf32, c) in enumerate(rows):
        s = np.abs(np.arange(2 * c + 1) - c).astype(np.float64)
        sig = float(st[k])
        pmf64 = norm.cdf((0.5 - s) / sig) - norm.cdf((-0.5 - s) / sig)
        tail = 2 * norm.cdf((-0.5 - s[0]) / sig)
        a = np.asarray(exact_pmf_to_quantized_cdf(list(pmf64.astype(np.float32)) + [np.float32(tail)]))
        b = np.asarray(exact_pmf_to_quantized_cdf(pmf32))
        worst = max(worst, int(np.abs(a - b).max()))
        differing += int((a != b).sum())
        total += len(a)
    assert worst <= 2 and differing <= 0.05 * total, (worst, differing, total)


def test_quantised_cdf_factorised_prior_rows_and_adversarial_pmfs():
    cq = _c_quantizer()
    sd = {k: seeded_tensor(k, s) for k, s in
          [(f"entropy_model_z._matrix{i}", (8, (3, 3, 3, 3, 1)[i], (1, 3, 3, 3, 3)[i])) for i in range(5)]
          + [(f"entropy_model_z._bias{i}", (8, (3, 3, 3, 3, 1)[i], 1)) for i in range(5)]
          + [(f"entropy_model_z._factor{i}", (8, 3, 1)) for i in range(4)] + [("entropy_model_z.quantiles", (8, 1, 3))]}
    table, lengths, offsets = O.eb_cdf_tables(sd)
    for c in range(8):
        row = table[c, : int(lengths[c])].tolist()
        assert row[0] == 0 and row[-1] == 65536 and all(b > a for a, b in zip(row, row[1:]))
    rng = np.random.default_rng(0)
    cases = [[1.0], [0.5, 0.5], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0], [1e-9] * 40 + [1.0], [2.0 ** -17] * 9 + [0.9],
             [0.25, 0.0, 0.25, 0.0, 0.5], list(rng.dirichlet(np.full(300, 0.05))), list(rng.dirichlet(np.full(7, 5.0))),
             [3e-6, 0.4999985, 0.4999985, 3e-6], list(np.full(1000, 1e-3))]
    for p in cases:
        ex = exact_pmf_to_quantized_cdf(p)
        assert ex == O.pmf_to_quantized_cdf(p) == cq(p), p[:6]
        assert ex[-1] == 65536 and all(b > a for a, b in zip(ex, ex[1:]))


def test_build_indexes_and_scale_table():
    st = O.get_scale_table()
    assert len(st) == 64 and abs(float(st[0]) - 0.11) < 1e-7 and abs(float(st[-1]) - 256.0) < 1e-3
    s = torch.tensor([-3.0, 0.0, 0.11, 0.1100001, float(st[1]), float(st[1]) * 1.0001, 255.0, 256.0, 1e6])
    idx = O.build_indexes(s).tolist()
    # index = number of table entries strictly below max(sigma, 0.11), capped at 63
    ref = [min(63, int((st < max(v, 0.11)).sum())) for v in s.tolist()]
    assert idx == ref, (idx, ref)
    assert idx[0] == idx[1] == idx[2] == 0 and idx[-1] == 63


# ---------------------------------------------------------------------------------------------------------- (iii)
def test_against_compressai_when_installed():
    """Auto-enabling pin: runs wherever compressai (ideally ==1.2.4) is importable."""
    compressai = pytest.importorskip("compressai")
    from compressai.entropy_models import EntropyBottleneck as CEB, GaussianConditional as CGC
    from compressai import ans
    from compressai._CXX import pmf_to_quantized_cdf as cxx_pmf
    # quantiser
    st, rows = _gaussian_rows()
    for pmf, _ in rows[::7]:
        assert list(cxx_pmf([float(np.float32(p)) for p in pmf], 16)) == O.pmf_to_quantized_cdf(pmf)
    # Gaussian conditional: tables, indexes, likelihood with both bounds
    gc = CGC(None)
    gc.update_scale_table(O.get_scale_table())
    table, lengths, offsets = O.gaussian_cdf_tables()
    assert gc._quantized_cdf.numpy().tolist() == table.tolist()
    assert gc._cdf_length.numpy().tolist() == lengths.tolist() and gc._offset.numpy().tolist() == offsets.tolist()
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2, 32, 6, 5, generator=g) * 6
    mu = torch.randn(2, 32, 6, 5, generator=g) * 4
    sg = torch.rand(2, 32, 6, 5, generator=g) * 3 - 0.5
    gc.eval()
    yh, lik = gc(y, sg, mu)
    ryh, rlik = O.gaussian_conditional(y, mu, sg, None)
    assert torch.equal(yh, ryh) and torch.allclose(lik, rlik, rtol=1e-6, atol=1e-9)
    assert torch.equal(gc.build_indexes(sg), O.build_indexes(sg).to(gc.build_indexes(sg).dtype))
    # rANS bytes
    sym = torch.round(y - mu).int()
    idx = O.build_indexes(sg)
    enc = ans.RansEncoder()
    theirs = enc.encode_with_indexes(sym.reshape(-1).tolist(), idx.reshape(-1).tolist(), table.tolist(), lengths.tolist(), offsets.tolist())
    ours = O.rans_encode(sym.reshape(-1).tolist(), idx.reshape(-1).tolist(), table, lengths, offsets)
    assert bytes(theirs) == bytes(ours)
    from crdr_amd.codec import rans
    assert bytes(theirs) == rans.encode_with_indexes(sym.reshape(-1).numpy(), idx.reshape(-1).int().numpy(), table, lengths, offsets)
    # factorised prior: parameters copied over by name (either naming scheme), then update() tables and likelihood
    C = 8
    eb = CEB(C)
    sd = {}
    for k, v in eb.state_dict().items():
        parts = k.split(".")
        name = k
        if parts[0] in ("matrices", "biases", "factors"):
            name = {"matrices": "_matrix", "biases": "_bias", "factors": "_factor"}[parts[0]] + parts[1]
        if name.startswith("_") and name[1:2].isalpha() and not name.startswith("_quantized") and not name.startswith("_offset") and not name.startswith("_cdf"):
            sd[k] = seeded_tensor("entropy_model_z." + name, v.shape)
        elif name == "quantiles":
            sd[k] = seeded_tensor("entropy_model_z.quantiles", v.shape)
        else:
            sd[k] = v
    eb.load_state_dict(sd)
    osd = {"entropy_model_z." + ({"matrices": "_matrix", "biases": "_bias", "factors": "_factor"}[k.split(".")[0]] + k.split(".")[1]
                                  if k.split(".")[0] in ("matrices", "biases", "factors") else k): v for k, v in sd.items()}
    eb.update(force=True)
    t, l, o = O.eb_cdf_tables(osd)
    assert eb._quantized_cdf.numpy().tolist() == t.tolist() and eb._cdf_length.numpy().tolist() == l.tolist() and eb._offset.numpy().tolist() == o.tolist()
    z = torch.randn(2, C, 4, 4, generator=g) * 3
    eb.eval()
    zh, zl = eb(z)
    rzh, rzl = O.entropy_bottleneck(osd, "entropy_model_z", z, None)
    assert torch.allclose(zh, rzh) and torch.allclose(zl, rzl, rtol=1e-5, atol=1e-9)
    assert abs(float(eb.loss()) - float(O.eb_aux_loss(osd, "entropy_model_z"))) < 1e-3


# ---------------------------------------------------------------------------------------------------------- (v)
def test_load_learned_weight_with_compressai_style_checkpoint(tmp_path):
    """base_model.py:80-118: DataParallel 'module.' prefix stripped, keys intersected with the model's own, CDF buffers
    take the checkpoint's shape, ParameterList-style factorised-prior names (compressai >= 1.2.5) accepted, and the
    factorised prior's tables are (re)built only if the checkpoint did not bring them."""
    from crdr_amd.models import build_comp_model
    from crdr_amd.utils.options import BaseConfig, ConfigDict
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", "_base_", "model", "beta_cond_interp_ca_elic_charm.yaml"))
    cfg["device"] = "cpu"
    torch.manual_seed(0)
    model = build_comp_model(ConfigDict(cfg))
    own = model.state_dict()
    ck = {}
    for k, v in own.items():
        if k.startswith("context_model.scale_slice_transforms.9"):
            continue  # a checkpoint from an earlier stage may lack keys: the model keeps its own values
        name = k
        parts = k.split(".")
        if parts[0] == "entropy_model_z":
            for old, new in (("_matrix", "matrices"), ("_bias", "biases"), ("_factor", "factors")):
                if parts[1].startswith(old):
                    name = f"entropy_model_z.{new}.{parts[1][len(old):]}"
        ck["module." + name] = torch.full_like(v, 0.5) if v.is_floating_point() and v.numel() else v
    ck["module.entropy_model_z._quantized_cdf"] = torch.arange(192 * 7, dtype=torch.int32).reshape(192, 7)
    ck["module.entropy_model_z._cdf_length"] = torch.full((192,), 7, dtype=torch.int32)
    ck["module.entropy_model_z._offset"] = torch.full((192,), -2, dtype=torch.int32)
    ck["module.entropy_model_y._quantized_cdf"] = torch.ones((64, 11), dtype=torch.int32)
    ck["module.entropy_model_y._cdf_length"] = torch.full((64,), 11, dtype=torch.int32)
    ck["module.entropy_model_y._offset"] = torch.full((64,), -4, dtype=torch.int32)
    ck["module.entropy_model_y.scale_table"] = torch.linspace(0.11, 256, 64)
    ck["module.some_other_net.weight"] = torch.zeros(3)  # not in the model: ignored
    path = tmp_path / "ck.pth.tar"
    torch.save({"comp_model": ck}, path)
    keep = own["context_model.scale_slice_transforms.9.model.0.weight"].clone()
    model.load_learned_weight(str(path))
    new = model.state_dict()
    assert torch.equal(new["context_model.scale_slice_transforms.9.model.0.weight"], keep)
    assert float(new["encoder.conv1.weight"].flatten()[0]) == 0.5 and float(new["entropy_model_z._matrix2"].flatten()[0]) == 0.5
    assert new["entropy_model_z._quantized_cdf"].shape == (192, 7) and int(new["entropy_model_z._offset"][0]) == -2
    assert new["entropy_model_y._quantized_cdf"].shape == (64, 11) and new["entropy_model_y.scale_table"].shape == (64,)
    # a checkpoint without tables: update(force=False) builds them
    for k in list(ck):
        if "_quantized_cdf" in k or "_cdf_length" in k or "_offset" in k:
            del ck[k]
    torch.save({"comp_model": ck}, path)
    model2 = build_comp_model(ConfigDict(cfg))
    model2.load_learned_weight(str(path))
    assert model2.entropy_model_z._quantized_cdf.numel() > 0 and model2.entropy_model_z._cdf_length.shape == (192,)


# ---------------------------------------------------------------------------------------------------------- GDN (oracle)
def test_gdn_oracle_known_answers():
    """at initialisation beta_eff = 1, gamma_eff = 0.1 I: y = x / sqrt(1 + 0.1 x^2); IGDN is its algebraic inverse on the
    norm; parameters below their bounds are clamped and receive gradient only when it would raise them."""
    b, g = O.gdn_init(8)
    sd = {"g.beta": b, "g.gamma": g}
    x = torch.linspace(-3, 3, 8 * 5).reshape(1, 8, 5, 1)
    y = O.gdn(sd, "g", x)
    assert torch.allclose(y, x / torch.sqrt(1 + 0.1 * x * x), atol=1e-6)
    yi = O.gdn(sd, "g", x, inverse=True)
    assert torch.allclose(yi, x * torch.sqrt(1 + 0.1 * x * x), atol=1e-5)
    # cross-channel mixing: gamma_eff = all 0.05 -> n = 1 + 0.05 * sum_j x_j^2
    ped = O.GDN_REPARAM_OFFSET ** 2
    sd2 = {"g.beta": b, "g.gamma": torch.sqrt(torch.full((8, 8), 0.05) + ped)}
    n = 1 + 0.05 * (x * x).sum(1, keepdim=True)
    assert torch.allclose(O.gdn(sd2, "g", x), x / torch.sqrt(n), atol=1e-6)
    # LowerBound rule on a parameter stored below its bound
    gam = torch.zeros(8, 8, requires_grad=True)  # < reparam_offset: clamped, gamma_eff = 0
    bet = b.clone().requires_grad_(True)
    out = O.gdn({"g.beta": bet, "g.gamma": gam}, "g", x)
    assert torch.allclose(out, x, atol=1e-6)
    out.sum().backward()
    assert float(gam.grad.abs().max()) >= 0.0 and torch.isfinite(gam.grad).all()
