"""Distortion metrics over a folder of reconstructions (the PSNR / LPIPS part of the reference's
scripts/calc_metrics.py:119-192; FID / KID / DISTS need third-party networks and are out of scope):

    python scripts/calc_metrics.py --real_dir kodak --fake_dir out [--metrics psnr lpips] [--lpips_weights alex.pth] [-d cuda:0]

PSNR follows the reference exactly: images are read back as uint8 RGB (what `imwrite` truncated to), the squared error is
taken in float32, PSNR is computed PER IMAGE (20 log10 255 - 10 log10 mse) and the per-image values are averaged
(calc_metrics.py:146,168).  LPIPS (AlexNet) runs on the GPU through the HIP LPIPS path used in training, on [-1, 1]
images, `lpips(fake, real)` per image, averaged."""
import argparse
import json
import os
import sys
from glob import glob

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def paired_paths(real_dir: str, fake_dir: str):
    real = sorted(glob(os.path.join(real_dir, "*.png")))
    fake = sorted(glob(os.path.join(fake_dir, "*.png")))
    assert len(real) == len(fake) and len(real) > 0, f"{len(real)} real vs {len(fake)} fake images"
    for r, f in zip(real, fake):
        assert os.path.basename(r) == os.path.basename(f), (r, f)
    return real, fake


def read_rgb_f32(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)


def psnr_one(real: np.ndarray, fake: np.ndarray) -> float:
    sqerror = np.sum(np.square(fake - real))
    mse = sqerror / real.size
    return float(20.0 * np.log10(255.0) - 10.0 * np.log10(mse))


def avg_psnr(real_paths, fake_paths) -> float:
    return float(np.mean([psnr_one(read_rgb_f32(r), read_rgb_f32(f)) for r, f in zip(real_paths, fake_paths)]))


def avg_lpips(real_paths, fake_paths, device: str, weights=None) -> float:
    import torch
    from crdr_amd.losses.perceptual_loss import LpipsAlex
    net = LpipsAlex().to(device)
    if weights:
        net.load_lpips_file(weights)  # {"alexnet_features": sd, "lpips_lin": sd}, see crdr_amd/losses/perceptual_loss.py
    else:
        print("warning: no --lpips_weights given, LPIPS runs on randomly initialised weights", file=sys.stderr)
    net.eval()
    vals = []
    with torch.no_grad():
        for r, f in zip(real_paths, fake_paths):
            a = torch.from_numpy(read_rgb_f32(r) / 255.0 * 2.0 - 1.0).permute(2, 0, 1).unsqueeze(0).to(device)
            b = torch.from_numpy(read_rgb_f32(f) / 255.0 * 2.0 - 1.0).permute(2, 0, 1).unsqueeze(0).to(device)
            vals.append(float(net(b, a).mean()))
    return float(np.mean(vals))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--real_dir", required=True)
    ap.add_argument("--fake_dir", required=True)
    ap.add_argument("--metrics", nargs="+", default=["psnr"], choices=["psnr", "lpips"])
    ap.add_argument("--lpips_weights", default=None)
    ap.add_argument("-d", "--device", default="cuda:0")
    ap.add_argument("--out", default=None, help="write the result dict as json")
    a = ap.parse_args(argv)
    real, fake = paired_paths(a.real_dir, a.fake_dir)
    res = {"num_images": len(real)}
    if "psnr" in a.metrics:
        res["PSNR"] = avg_psnr(real, fake)
    if "lpips" in a.metrics:
        res["LPIPS"] = avg_lpips(real, fake, a.device, a.lpips_weights)
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f)
    return res


if __name__ == "__main__":
    main()
