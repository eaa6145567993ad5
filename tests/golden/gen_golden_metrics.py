"""Golden vectors for the image-metric helpers, recorded from the REFERENCE's own functions (run once, in the build
container only; needs /root/reference):  src/utils/img_utils.py `calc_psnr` (:102-132: uint8 TRUNCATION of both images,
float32 mean of squared differences) and `torch2npimg` (:17-42).  Only seeds -> numbers are stored, no reference source.

    python tests/golden/gen_golden_metrics.py     # writes tests/golden/reference_metrics.json"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from gen_golden import install_stubs  # noqa: E402
from seeded_weights import seeded_input  # noqa: E402

CASES = [("a", (1, 3, 37, 53), 0.02), ("b", (1, 3, 64, 64), 0.1), ("c", (1, 3, 128, 96), 0.004), ("d", (1, 3, 16, 16), 0.5)]


def images(tag, shape, noise):
    real = seeded_input("psnr.real." + tag, shape)
    fake = (real + noise * seeded_input("psnr.noise." + tag, shape)).clamp(-1, 1)
    return real, fake


def main():
    install_stubs()
    sys.path.insert(0, "/root/reference")
    from src.utils import img_utils as R
    out = {}
    for tag, shape, noise in CASES:
        real, fake = images(tag, shape, noise)
        img = R.torch2npimg(fake)
        out[tag] = {"shape": list(shape), "noise": noise, "psnr": float(R.calc_psnr(real, fake, 255)),
                    "npimg_sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(), "npimg_shape": list(img.shape),
                    "npimg_sum": int(img.astype(np.int64).sum())}
    with open(os.path.join(HERE, "reference_metrics.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
