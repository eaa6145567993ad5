"""Compression / decompression entry point, argument compatible with the reference's scripts/compress.py:35-47:

    python scripts/compress.py --config_path config/crdr.yaml --model_path crdr.pth.tar --img_dir ./demo_images \
        --save_dir out -q 0.0 -b 3.84 --decompress -d cuda:0

Per image: <name>.bin (u32-LE length-prefixed header / z / y strings), optionally <name>.png; `_bitrates.csv` and
`_avg_bitrate.json` (avg of real_bpp = 8 * file bytes / (H W)).  Transforms, hyper-decoder and Charm run on the GPU
through the HIP kernels (deterministic), rANS on the host."""
import argparse
import json
import os
import sys
from glob import glob

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402

from crdr_amd.models import build_comp_model  # noqa: E402
from crdr_amd.utils import img_utils  # noqa: E402
from crdr_amd.utils.codec_utils import load_byte_strings, save_byte_strings  # noqa: E402
from crdr_amd.utils.logger import get_root_logger  # noqa: E402
from crdr_amd.utils.options import BaseConfig  # noqa: E402


class CustomConfig(BaseConfig):
    @classmethod
    def get_opt(cls, argv=None) -> "CustomConfig":
        args = cls.arg_parse(argv)
        cfg, text, _ = cls._file2dict_yaml(args["config_path"])
        opt = cls._merge_a_into_b(args, cfg)
        opt["is_train"] = False
        return cls(opt, cfg_text=text, filename=args["config_path"])

    @staticmethod
    def arg_parse(argv=None):
        ap = argparse.ArgumentParser()
        ap.add_argument("--config_path", type=str, help="path to .yaml")
        ap.add_argument("--model_path", type=str, help="path to model (.pth.tar)")
        ap.add_argument("--img_dir", type=str)
        ap.add_argument("--save_dir", type=str)
        ap.add_argument("-q", "--quality", type=float)
        ap.add_argument("-b", "--beta", type=float)
        ap.add_argument("--decompress", action="store_true")
        ap.add_argument("-d", "--device", type=str, default="cuda:0")
        return vars(ap.parse_args(argv))


def load_image(path: str) -> torch.Tensor:
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).float() / 255.0
    return ((t - 0.5) / 0.5).unsqueeze(0)


def main(argv=None):
    opt = CustomConfig.get_opt(argv)
    logger = get_root_logger()
    os.makedirs(opt.save_dir, exist_ok=True)
    paths = sorted(glob(os.path.join(opt.img_dir, "*.png")))
    ckw = {"rate_ind": opt.quality} if opt.quality is not None and opt.quality >= 0.0 else {}
    model = build_comp_model(opt).to(opt.device)
    if opt.get("model_path"):
        model.load_learned_weight(ckpt_path=opt.model_path)
    model.eval()
    model.codec_setup()
    rows = []
    for p in paths:
        name = os.path.basename(p)
        img = load_image(p)
        _, _, H, W = img.shape
        out = model.compress(img, **ckw)
        strings = out["string_list"]
        bin_path = os.path.join(opt.save_dir, name.replace(".png", ".bin"))
        save_byte_strings(bin_path, strings)
        nbytes = os.path.getsize(bin_path)
        rows.append({"img_name": name, "header_bit": len(strings[0]) * 8, "z_bit": len(strings[1]) * 8, "y_bit": len(strings[2]) * 8,
                     "real_bit": nbytes * 8, "real_bpp": nbytes * 8 / H / W, "pred_z_bit": out["pred_z_bit"],
                     "pred_y_bit": out["pred_y_bit"], "pred_bit": out["pred_z_bit"] + out["pred_y_bit"],
                     "pred_bpp": out["pred_z_bpp"] + out["pred_y_bpp"], "num_pixel": H * W})
        if opt.decompress:
            dkw = {"beta": opt.beta} if opt.beta is not None and opt.beta >= 0.0 else {}
            fake, z_hat, y_hat = model.decompress(load_byte_strings(bin_path), **dkw)
            img_utils.imwrite(os.path.join(opt.save_dir, name), fake)
    df = pd.json_normalize(rows)
    df.to_csv(os.path.join(opt.save_dir, "_bitrates.csv"))
    avg = float(df["real_bpp"].mean()) if len(df) else float("nan")
    with open(os.path.join(opt.save_dir, "_avg_bitrate.json"), "w") as f:
        json.dump({"avg_bpp": avg}, f)
    logger.warning(f"quality: {opt.quality}, beta: {opt.beta}; num_image: {len(paths)}; avg_bpp: {avg:.4f} [bpp]")
    return avg


if __name__ == "__main__":
    main()
