"""End-to-end codec parity of the HIP path against the ORACLE (not against itself): `model.compress(x, q)` through the HIP
transforms + C rANS must produce the header, z string and y string the oracle's compress() produces byte for byte
(oracle/crdr_oracle.py: restatement of interpca_hyperprior_charm_model.py:83-149 + minnen20_charm_context_model.py:143-240 on
its own CDF tables and Python coder), and the oracle's decompress() of the HIP bytes must land on the HIP encoder's y_hat / z_hat.

Two discontinuities separate any two correct fp32 implementations: round(y - mu) at k + 1/2 and the CDF index at a scale-table
entry.  The device's decisions are handed to the oracle ONLY inside the oracle's windows (FORCE_TOL 5e-4 absolute, INDEX_TOL 1e-4
relative); everywhere else they must agree exactly and the adopted ones are bounded to 0.1 % (check_forced / check_forced_indexes)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.seeded_weights import seeded_input
from tests.test_gpu_model import _full_model, close, dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kodak(name="kodim23.png"):
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(ROOT, "demo_images", name)).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0) * 2 - 1


@pytest.fixture(scope="module")
def setup():
    from oracle import crdr_oracle as O
    model, sd = _full_model(True)
    model.eval()
    model.codec_setup()
    return model, sd, O.codec_tables(sd)


def _device_decisions(ticket):
    sym, idx, z_sym, _ = ticket["_keep"]
    return {"y": sym.cpu(), "idx": idx.cpu(), "z": z_sym.cpu()}


CASES = [("seeded 64x64", (64, 64), 0.0), ("seeded 70x90", (70, 90), 2.25), ("seeded 128x64", (128, 64), 4.0), ("kodim23 768x512", None, 2.25)]


@pytest.mark.parametrize("name,size,q", CASES, ids=[c[0] for c in CASES])
def test_hip_compress_bytes_equal_oracle_bytes(setup, name, size, q):
    from oracle import crdr_oracle as O
    model, sd, tables = setup
    x = _kodak() if size is None else seeded_input(f"codec{size}", (1, 3, *size))
    ticket = model.compress_device(x, rate_ind=q)
    forced = _device_decisions(ticket)
    out = model.compress_finish(ticket)
    rep = {}
    ref = O.compress(sd, x, q, tables, forced=forced, report=rep)
    O.check_forced(rep, rep["symbols"])
    O.check_forced_indexes(rep)
    # the device tables are the oracle's tables
    cdf, sizes, offs = model.entropy_model_y.host_tables()
    assert np.array_equal(np.asarray(cdf), np.asarray(tables["y"][0])) and list(sizes) == tables["y"][1] and list(offs) == tables["y"][2]
    # symbols, indexes, then the three byte strings
    assert torch.equal(forced["z"].reshape(-1).int(), ref["z_symbols"].reshape(-1))
    assert torch.equal(forced["y"].reshape(-1).int(), ref["y_symbols"].reshape(-1))
    assert torch.equal(forced["idx"].reshape(-1).int(), ref["indexes"].reshape(-1).int())
    hdr, zs, ys = out["string_list"]
    assert hdr == ref["string_list"][0], (hdr.hex(), ref["string_list"][0].hex())
    assert zs == ref["string_list"][1], "z string differs from the oracle's"
    assert ys == ref["string_list"][2], "y string differs from the oracle's"
    close(out["y_hat"], ref["y_hat"], "y_hat", 3e-4)
    close(out["z_hat"], ref["z_hat"], "z_hat", 1e-6)
    assert abs(out["pred_y_bit"] - ref["pred_y_bit"]) <= 2e-4 * ref["pred_y_bit"] and abs(out["pred_z_bit"] - ref["pred_z_bit"]) <= 2e-4 * ref["pred_z_bit"]
    # information-theoretic accounting of the HIP bytes on the oracle's tables
    ideal = O.ideal_code_length_bits(ref["y_symbols"].reshape(-1).tolist(), ref["indexes"].reshape(-1).tolist(), tables["y"])
    assert ideal - 32 <= 8 * len(ys) <= ideal + 64, (8 * len(ys), ideal)
    # the ORACLE decodes the HIP bytes (CDF indexes adopted only at table entries) to the HIP encoder's symbols and latents
    rep2 = {}
    dec = O.decompress(sd, out["string_list"], 3.84, tables, forced={"idx": forced["idx"]}, report=rep2)
    O.check_forced_indexes(rep2)
    assert torch.equal(dec["y_symbols"].reshape(-1), forced["y"].reshape(-1).int()) and torch.equal(dec["z_symbols"].reshape(-1), forced["z"].reshape(-1).int())
    close(out["y_hat"], dec["y_hat"], "oracle-decoded y_hat", 3e-4)
    assert torch.equal(dec["z_hat"], out["z_hat"].cpu().reshape(dec["z_hat"].shape))
    # and the HIP decoder's image is the oracle decoder's image
    fake, z_hat, y_hat = model.decompress(out["string_list"], beta=3.84)
    assert torch.equal(y_hat.cpu(), out["y_hat"].cpu())
    close(fake, dec["fake_images"], "decoded image", 1e-3)


def test_eval_forward_at_kodak_size_matches_oracle(setup):
    """run_model(is_train=False) on a 768x512 Kodak image: x_hat, bpp, qbpp against the oracle (a Kodak-size shape meets the
    oracle, not only itself)."""
    from oracle import crdr_oracle as O
    model, sd, _ = setup
    x = _kodak("kodim03.png")
    model.context_model.record_symbols = []
    out = model.run_model(x, rate_ind=1.5, beta=2.56, is_train=False)
    forced = {"y": [t.cpu() for t in model.context_model.record_symbols],
              "z": torch.round(out["z_hat"].detach().cpu() - sd["entropy_model_z.quantiles"][:, 0, 1].reshape(1, -1, 1, 1))}
    model.context_model.record_symbols = None
    rep = {}
    with torch.no_grad():
        ref = O.generator_forward(sd, x, 1.5, 2.56, is_train=False, forced=forced, report=rep)
    O.check_forced(rep, rep["symbols"])
    close(out["fake_images"], ref["fake_images"], "x_hat 768x512", 1e-3)
    close(out["bpp"], ref["qbpp"], "bpp (eval: quantised likelihood)", 1e-4)
    close(out["qbpp"], ref["qbpp"], "qbpp", 1e-4)
