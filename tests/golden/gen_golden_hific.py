"""Golden vectors of the reference's HiFiC discriminators, same rules as gen_golden.py: the
reference's own modules (src/models/discriminator/hific_discriminator.py:10-58, torch.nn.utils.spectral_norm) are run
on seeded weights; only inputs -> outputs are stored.

    python tests/golden/gen_golden_hific.py      # needs /root/reference; writes tests/golden/reference_hific.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from gen_golden import REF, install_stubs, npy  # noqa: E402
from seeded_weights import fill_module_, seeded_input  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import logging
    logging.disable(logging.CRITICAL)
    from src.models.discriminator.hific_discriminator import HiFiCConditionalDiscriminator, HiFiCDiscriminator
    out = {}
    x = seeded_input("hific.x", (2, 3, 32, 48))
    y = seeded_input("hific.y", (2, 24, 2, 3))
    out["in.x"], out["in.y"] = npy(x), npy(y)
    for name, D, kw in (("plain", HiFiCDiscriminator(in_ch=3, out_ch=1, main_ch=16, use_sn=True), {}),
                        ("nosn", HiFiCDiscriminator(in_ch=3, out_ch=1, main_ch=16, use_sn=False), {}),
                        ("cond", HiFiCConditionalDiscriminator(in_ch=3, out_ch=1, main_ch=16, y_ch=24, latent_nc=4, use_sn=True), {"y_hat": y})):
        fill_module_(D, f"hific.{name}.")
        D.eval()
        out[f"{name}.eval"] = npy(D(x, **kw))
        D.train()
        xi = x.clone().requires_grad_(True)
        o = D(xi, **kw)            # one power iteration per spectral-normed layer, u / v updated in place
        gy = seeded_input(f"hific.{name}.gy", tuple(o.shape))
        o.backward(gy)
        out[f"{name}.train"] = npy(o)
        out[f"{name}.train.dx"] = npy(xi.grad)
        sd = D.state_dict()
        for k in sd:
            if k.endswith("weight_u") or k.endswith("weight_v"):
                out[f"{name}.after.{k}"] = npy(sd[k])
        for k, p in D.named_parameters():
            if k in ("model.0.weight_orig", "model.6.weight_orig", "model.8.weight_orig", "model.0.weight", "model.8.bias", "latent_conv.0.weight"):
                g = p.grad
                out[f"{name}.grad.{k}"] = npy(g[:, :8] if g.numel() > 20000 else g)  # big tensors: first 8 input channels
        out[f"{name}.keys"] = np.array(sorted(sd.keys()))
    np.savez_compressed(os.path.join(HERE, "reference_hific.npz"), **out)
    print({k: v.shape for k, v in out.items() if not k.endswith("keys")})


if __name__ == "__main__":
    main()
