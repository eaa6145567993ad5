"""(q, beta) grid evaluation in the layout of the reference's rd_results/*.csv (README.md:52-58,76; rd_results/README.md):

    python scripts/eval_grid.py --config_path config/crdr.yaml --model_path crdr.pth.tar --img_dir ./kodak \
        --dataset kodak --out_csv rd_results/kodak.csv [--qualities 0 0.25 ... 4] [--betas 3.84 0] [--lpips_weights f]

For every quality q (default 0, 0.25, ..., 4: 17 levels) each image is compressed ONCE -- the bitstream does not depend on
beta (kodak.csv rows beta = 0 and 3.84 carry identical bpp) -- and decoded once per beta.  Per (q, beta) row:
`dataset,quality,beta,bpp,PSNR,LPIPS,DISTS` with bpp = mean over images of 8 * bytes(.bin) / (H W) (scripts/compress.py:
108-121), PSNR = mean over images of the per-image PSNR of the uint8-truncated PNGs (scripts/calc_metrics.py:119-168),
LPIPS through the HIP LPIPS-Alex when weights are supplied (else empty), DISTS empty (third-party network, out of scope).
With `--reference_csv rd_results/kodak.csv` the result is compared row by row with the published numbers (bpp +-1e-4,
PSNR +-0.01 dB: BASELINE.json's gate) -- meaningful once the pretrained checkpoint is supplied."""
import argparse
import csv
import json
import os
import sys
import tempfile
from glob import glob

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

COLUMNS = ["dataset", "quality", "beta", "bpp", "PSNR", "LPIPS", "DISTS"]


def default_qualities():
    return [round(0.25 * i, 2) for i in range(17)]


def evaluate(model, img_paths, qualities, betas, work_dir, lpips_net=None, dataset="kodak", log=print):
    """-> list of row dicts in (beta-major, quality-minor) order like the reference's csv files."""
    from crdr_amd.utils import img_utils
    from crdr_amd.utils.codec_utils import load_byte_strings, save_byte_strings
    from scripts.calc_metrics import psnr_one, read_rgb_f32
    from scripts.compress import load_image
    cells = {}
    for q in qualities:
        per_beta = {b: {"bpp": [], "psnr": [], "lpips": []} for b in betas}
        for p in img_paths:
            img = load_image(p)
            _, _, H, W = img.shape
            out = model.compress(img, rate_ind=q)
            bin_path = os.path.join(work_dir, "tmp.bin")
            save_byte_strings(bin_path, out["string_list"])
            bpp = os.path.getsize(bin_path) * 8 / H / W
            strings = load_byte_strings(bin_path)
            for b in betas:
                fake, _, _ = model.decompress(strings, beta=b)
                png = os.path.join(work_dir, "tmp.png")
                img_utils.imwrite(png, fake)
                c = per_beta[b]
                c["bpp"].append(bpp)
                c["psnr"].append(psnr_one(read_rgb_f32(p), read_rgb_f32(png)))
                if lpips_net is not None:
                    with torch.no_grad():
                        a = torch.from_numpy(read_rgb_f32(p) / 255.0 * 2.0 - 1.0).permute(2, 0, 1).unsqueeze(0).to(fake.device)
                        f = torch.from_numpy(read_rgb_f32(png) / 255.0 * 2.0 - 1.0).permute(2, 0, 1).unsqueeze(0).to(fake.device)
                        c["lpips"].append(float(lpips_net(f, a).mean()))
        for b in betas:
            c = per_beta[b]
            cells[(q, b)] = {"dataset": dataset, "quality": q, "beta": b, "bpp": float(np.mean(c["bpp"])), "PSNR": float(np.mean(c["psnr"])),
                             "LPIPS": float(np.mean(c["lpips"])) if c["lpips"] else "", "DISTS": ""}
            log(f"q={q} beta={b}: bpp {cells[(q, b)]['bpp']:.6f} PSNR {cells[(q, b)]['PSNR']:.4f}")
    return [cells[(q, b)] for b in betas for q in qualities]


def write_csv(rows, path):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=COLUMNS)
        w.writeheader()
        for r in rows:
            w.writerow(r)


def compare_with_reference(rows, ref_csv, bpp_tol=1e-4, psnr_tol=0.01):
    """-> list of (quality, beta, d_bpp, d_psnr, ok) for the rows present in both."""
    ref = {}
    with open(ref_csv) as f:
        for r in csv.DictReader(f):
            ref[(round(float(r["quality"]), 4), round(float(r["beta"]), 4))] = r
    out = []
    for r in rows:
        k = (round(float(r["quality"]), 4), round(float(r["beta"]), 4))
        if k in ref:
            db, dp = r["bpp"] - float(ref[k]["bpp"]), r["PSNR"] - float(ref[k]["PSNR"])
            out.append((k[0], k[1], db, dp, abs(db) <= bpp_tol and abs(dp) <= psnr_tol))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", required=True)
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--img_dir", required=True)
    ap.add_argument("--dataset", default="kodak")
    ap.add_argument("--out_csv", required=True)
    ap.add_argument("--qualities", type=float, nargs="+", default=None)
    ap.add_argument("--betas", type=float, nargs="+", default=[3.84, 0.0])
    ap.add_argument("--lpips_weights", default=None)
    ap.add_argument("--reference_csv", default=None)
    ap.add_argument("-d", "--device", default="cuda:0")
    a = ap.parse_args(argv)
    from crdr_amd.models import build_comp_model
    from crdr_amd.utils.options import BaseConfig
    cfg, text, _ = BaseConfig._file2dict_yaml(a.config_path)
    cfg["device"], cfg["is_train"] = a.device, False
    model = build_comp_model(BaseConfig(cfg, cfg_text=text, filename=a.config_path)).to(a.device)
    if a.model_path:
        model.load_learned_weight(ckpt_path=a.model_path)
    else:
        print("warning: no --model_path: evaluating RANDOM weights (plumbing check only)", file=sys.stderr)
    model.eval()
    model.codec_setup()
    lp = None
    if a.lpips_weights:
        from crdr_amd.losses.perceptual_loss import LpipsAlex
        lp = LpipsAlex().to(a.device).eval()
        lp.load_lpips_file(a.lpips_weights)
    paths = sorted(glob(os.path.join(a.img_dir, "*.png")))
    assert paths, f"no .png images under {a.img_dir}"
    with tempfile.TemporaryDirectory() as tmp:
        rows = evaluate(model, paths, a.qualities or default_qualities(), a.betas, tmp, lp, a.dataset)
    os.makedirs(os.path.dirname(os.path.abspath(a.out_csv)), exist_ok=True)
    write_csv(rows, a.out_csv)
    res = {"rows": len(rows), "images": len(paths), "out_csv": a.out_csv}
    if a.reference_csv:
        cmp = compare_with_reference(rows, a.reference_csv)
        res["compared"] = len(cmp)
        res["within_gate"] = sum(1 for c in cmp if c[4])
        res["max_abs_dbpp"] = max((abs(c[2]) for c in cmp), default=None)
        res["max_abs_dpsnr"] = max((abs(c[3]) for c in cmp), default=None)
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
