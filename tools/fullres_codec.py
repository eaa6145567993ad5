"""BASELINE config #5 stand-in: compress -> decompress of one CLIC-sized image (seeded smooth noise, random-init weights)
on one GPU; prints the wall-time split and checks the decoder reproduces y_hat bit for bit.
Usage: python tools/fullres_codec.py [H W] [q] [beta]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.models import build_comp_model  # noqa: E402
from crdr_amd.utils.options import BaseConfig, ConfigDict  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1365, 2048)
    q = float(sys.argv[3]) if len(sys.argv) > 3 else 2.25
    beta = float(sys.argv[4]) if len(sys.argv) > 4 else 3.84
    cfg, _, _ = BaseConfig._file2dict_yaml(os.path.join(ROOT, "config", "crdr.yaml"))
    cfg["device"], cfg["is_train"] = "cuda:0", False
    torch.manual_seed(0)
    t = time.perf_counter()
    model = build_comp_model(ConfigDict(cfg)).to("cuda:0").eval()
    print("build", round(time.perf_counter() - t, 2), flush=True); t = time.perf_counter()
    model.codec_setup()
    print("codec_setup", round(time.perf_counter() - t, 2), flush=True)
    g = torch.Generator().manual_seed(1)
    small = torch.rand(1, 3, h // 16 + 1, w // 16 + 1, generator=g) * 2 - 1
    x = torch.nn.functional.interpolate(small, size=(h, w), mode="bicubic", align_corners=False).clamp(-1, 1)
    for rep in range(2):  # second pass: warm (packs, workspaces, tuned paths)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.compress(x, rate_ind=q)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        fake, z_hat, y_hat = model.decompress(out["string_list"], beta=beta)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("pass", rep, "compress", round(t1 - t0, 3), "decompress", round(t2 - t1, 3), flush=True)
    nbytes = sum(len(s) for s in out["string_list"]) + 12
    ok = torch.equal(y_hat.cpu(), out["y_hat"].cpu()) and torch.equal(z_hat.cpu(), out["z_hat"].cpu())
    print({"size": (h, w), "q": q, "beta": beta, "compress_s": round(t1 - t0, 3), "decompress_s": round(t2 - t1, 3),
           "bpp": round(nbytes * 8 / h / w, 4), "roundtrip_bit_exact": ok, "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)})
    assert ok and fake.shape == (1, 3, h, w)


if __name__ == "__main__":
    main()
