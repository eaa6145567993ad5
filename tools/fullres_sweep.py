"""BASELINE config #5 at its stated grid, on stand-ins: the reference's evaluation sweep (scripts/compress.py:70-138 driven over
q in {0, 0.25, ..., 4.0} x beta in {0, 3.84}, rd_results/README.md:3-17) over K synthetic CLIC-sized images (seeded smooth noise,
2048 x 1365; the CLIC2020 test set is not in the container) with random-init weights.

bpp depends on q only (rd_results/kodak.csv rows beta = 0 vs 3.84 are identical), so the sweep is 17 encodes + 34 decodes per image:
  * encode: `compress_many` (host rANS coder of image k in worker threads beside the GPU work of image k + 1) per q,
  * decode: `decompress_many` per (q, beta),
each checked against the serial form (`compress` / `decompress` per image) on the first `--check` rate points: identical bytes,
bit-identical y_hat / z_hat / image.  Prints one JSON document: totals, per-q bpp, the wall split {transforms, Charm, rANS} of the
serial form summed over the sweep (model.codec_profile), pipelined / serial ratios.

    python tools/fullres_sweep.py [--images 16] [--size 1365 2048] [--check 3] > profiles/r5_fullres_sweep.json"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.models import build_comp_model  # noqa: E402
from crdr_amd.utils.options import BaseConfig, ConfigDict  # noqa: E402
from tools.fullres_codec import smooth_image  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def seed_interp_ca(model) -> None:
    """Default-init InterpChAtt is the identity at every rate level (W = ln(e - 1), b = 0: interp_channel_attention.py:39-46), so a sweep over
    q on a freshly built model codes the SAME latents 17 times (round 4: bpp 0.5377 for every q).  Seeded, non-identity weights instead:
    on the analysis side raw weights that grow strictly with the level in every channel (per-channel offset + per-channel positive slope,
    tests/golden/seeded_weights.py generators), so that softplus(lerp(W[floor q], W[ceil q])) -- the lerp comes BEFORE the softplus,
    interp_channel_attention.py:47-52 -- scales every stage up as q grows and the rate must grow with it; biases seeded but level
    independent; on the synthesis side seeded weights of every level (they do not touch the bitstream)."""
    import math
    from tests.golden.seeded_weights import seeded_tensor
    base = math.log(math.e - 1)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if "interp_ca_list" not in name:
                continue
            L = prm.shape[0]
            noise = seeded_tensor(name, prm.shape).to(prm.device) - (base if name.endswith("weight") else 0.0)   # 0.3 N(0, 1), [L,1,C,1,1]
            if name.startswith("encoder"):
                lvl = torch.arange(L, device=prm.device, dtype=prm.dtype).view(L, 1, 1, 1, 1)
                if name.endswith("weight"):
                    # nine stages multiply: ~0.97^9 of the identity's latent scale at q = 0 (still codes non-zero symbols), ~1.09^9 at q = 4
                    off, slope = 0.3 * noise[:1], 0.03 + 0.03 * torch.sigmoid(4.0 * noise[1:2])     # slope in (0.03, 0.06) per channel and level
                    prm.copy_(base + off + (lvl - 1.0) * slope)
                else:
                    prm.copy_((0.1 * noise[:1]).expand_as(prm))
            else:
                prm.copy_((base if name.endswith("weight") else 0.0) + noise)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--size", type=int, nargs=2, default=(1365, 2048))
    ap.add_argument("--check", type=int, default=3, help="rate points (spread over the grid) on which the serial form is run too")
    ap.add_argument("--workers", type=int, default=2)
    a = ap.parse_args()
    h, w = a.size
    qs = [0.25 * i for i in range(17)]
    betas = [0.0, 3.84]
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", "crdr.yaml"))
    cfg["device"], cfg["is_train"] = "cuda:0", False
    torch.manual_seed(0)
    model = build_comp_model(ConfigDict(cfg)).to("cuda:0").eval()
    seed_interp_ca(model)
    model.codec_setup()
    imgs = [smooth_image(h, w, 1 + k) for k in range(a.images)]
    out = model.compress(imgs[0], rate_ind=2.0)          # warm-up: packs, workspaces, pinned buffers
    model.decompress(out["string_list"], beta=3.84)
    torch.cuda.synchronize()
    check_q = sorted({qs[round(i * (len(qs) - 1) / max(a.check - 1, 1))] for i in range(a.check)}) if a.check else []

    def sync_time():
        torch.cuda.synchronize()
        return time.perf_counter()
    t_enc = t_dec = 0.0
    ser = {"enc_s": 0.0, "dec_s": 0.0, "enc_split": {}, "dec_split": {}, "points": 0}
    bpp, streams = {}, {}
    for q in qs:
        t = sync_time()
        sl = [o["string_list"] for o in model.compress_many(imgs, workers=a.workers, rate_ind=q)]
        t_enc += sync_time() - t
        streams[q] = sl
        bpp[q] = sum(8.0 * (sum(len(s) for s in one) + 12) / (h * w) for one in sl) / len(sl)
        for one in sl:   # the header's rate byte is floor(16 q) (codec_utils.py:82-103)
            assert one[0][5] == int(16 * q), (q, one[0][5])
        if q in check_q:   # the serial form on the same images: identical bytes, and its wall split
            model.codec_profile = {}
            t = sync_time()
            ref = [model.compress(im, rate_ind=q)["string_list"] for im in imgs]
            ser["enc_s"] += sync_time() - t
            for k, v in model.codec_profile.items():
                ser["enc_split"][k] = ser["enc_split"].get(k, 0.0) + v
            model.codec_profile = None
            assert ref == sl, f"q = {q}: the pipelined encode produced different bytes"
            ser["points"] += 1
    dec_checked = 0
    for q in qs:
        for beta in betas:
            t = sync_time()
            outs = list(model.decompress_many(streams[q], workers=a.workers + 1, beta=beta))
            t_dec += sync_time() - t
            if q in check_q:
                model.codec_profile = {}
                t = sync_time()
                ref = [model.decompress(sl, beta=beta) for sl in streams[q]]
                ser["dec_s"] += sync_time() - t
                for k, v in model.codec_profile.items():
                    ser["dec_split"][k] = ser["dec_split"].get(k, 0.0) + v
                model.codec_profile = None
                for (f0, z0, y0), (f1, z1, y1) in zip(ref, outs):
                    assert torch.equal(f0, f1) and torch.equal(z0, z1) and torch.equal(y0, y1), f"q = {q}, beta = {beta}: the pipelined decode differs"
                dec_checked += 1
            del outs
    # the q-dependence of the codec path is live: 17 different bitstreams per image, bpp strictly increasing in q; beta never reaches the
    # encoder (compress(real_images, rate_ind): beta_cond_interpca_hyperprior_charm_model.py:84-118; rd_results/README.md:3), so the two
    # decodes of a rate point share one stream by construction
    import hashlib
    digests = {q: [hashlib.sha256(b"".join(one)).hexdigest() for one in streams[q]] for q in qs}
    distinct = min(len({digests[q][k] for q in qs}) for k in range(a.images))
    assert distinct == len(qs), f"only {distinct} distinct bitstreams over {len(qs)} rate points"
    strictly = all(bpp[qs[i]] < bpp[qs[i + 1]] for i in range(len(qs) - 1))
    assert strictly, ("bpp is not strictly increasing in q", bpp)
    n_enc, n_dec = len(qs) * a.images, len(qs) * len(betas) * a.images
    doc = {"what": "BASELINE config #5 stand-in: 17 q x 2 beta over synthetic 2048x1365 images, random-init weights (bpp is a property of the "
                   "random model, not of CRDR); compress_many / decompress_many, checked against the serial form on `checked_q`",
           "images": a.images, "size": [h, w], "q_grid": qs, "betas": betas,
           "encodes": n_enc, "decodes": n_dec,
           "pipelined": {"encode_s": round(t_enc, 3), "decode_s": round(t_dec, 3), "ms_per_encode": round(1e3 * t_enc / n_enc, 2),
                         "ms_per_decode": round(1e3 * t_dec / n_dec, 2), "total_s": round(t_enc + t_dec, 3),
                         "images_per_s_full_grid": round(a.images / (t_enc + t_dec), 3)},
           "checked_q": check_q, "bytes_identical_serial_vs_pipelined": True, "decodes_bit_identical_serial_vs_pipelined": dec_checked,
           "serial_on_checked_points": {
               "ms_per_encode": round(1e3 * ser["enc_s"] / max(ser["points"] * a.images, 1), 2),
               "ms_per_decode": round(1e3 * ser["dec_s"] / max(dec_checked * a.images, 1), 2),
               "encode_split_ms_per_image": {k: round(1e3 * v / max(ser["points"] * a.images, 1), 2) for k, v in ser["enc_split"].items()},
               "decode_split_ms_per_image": {k: round(1e3 * v / max(dec_checked * a.images, 1), 2) for k, v in ser["dec_split"].items()}},
           "bpp_by_q": {f"{q:.2f}": round(v, 4) for q, v in bpp.items()},
           "bpp_strictly_increasing_in_q": strictly, "distinct_bitstreams_per_image": distinct,
           "interp_ca": "seeded non-identity weights (seed_interp_ca): analysis side strictly increasing with the level in every channel",
           "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
