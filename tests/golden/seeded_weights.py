"""Deterministic, construction-order-independent weights keyed by state-dict name.

Used three ways with identical results: (1) tests/golden/gen_golden.py fills the *reference* modules before
recording their outputs, (2) the oracle tests fill a plain dict, (3) the GPU tests fill the HIP modules.
Because values depend only on (key, shape), a key or shape mismatch between the implementations is an error,
which pins the checkpoint schema as a side effect.
"""
import math
import zlib

import torch


def seeded_tensor(key: str, shape, salt: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * salt) % (2 ** 31))
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if "interp_ca_list" in key:  # [L,1,C,1,1]: around the identity init ln(e-1) / 0
        base = math.log(math.e - 1) if leaf == "weight" else 0.0
        return base + 0.3 * torch.randn(shape, generator=g)
    if leaf == "quantiles":  # [C,1,3] ordered triples
        c = shape[0]
        med = 0.5 * torch.randn(c, generator=g)
        w = 6.0 + 4.0 * torch.rand(c, 2, generator=g)
        return torch.stack([med - w[:, 0], med, med + w[:, 1]], 1).reshape(shape)
    if leaf.startswith("_matrix"):
        return 0.5 * torch.randn(shape, generator=g)
    if leaf.startswith("_bias"):
        return torch.rand(shape, generator=g) - 0.5
    if leaf.startswith("_factor"):
        return 0.3 * torch.randn(shape, generator=g)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    return torch.randn(shape, generator=g)


def fill_state_dict(shapes: dict, salt: int = 0) -> dict:
    """shapes: {key: shape} -> {key: tensor}"""
    return {k: seeded_tensor(k, s, salt) for k, s in shapes.items()}


def fill_module_(module: torch.nn.Module, prefix: str = "", salt: int = 0) -> dict:
    """Overwrite every floating parameter/buffer of `module` in place; returns the {prefixed key: tensor} dict."""
    out = {}
    sd = module.state_dict()
    for k, v in sd.items():
        if not torch.is_floating_point(v) or v.numel() == 0:
            continue
        t = seeded_tensor(prefix + k, v.shape, salt).to(v.dtype)
        out[prefix + k] = t
        sd[k] = t
    module.load_state_dict(sd)
    return out


def seeded_input(tag: str, shape, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(("input:" + tag).encode()) % (2 ** 31))
    return (torch.rand(tuple(shape), generator=g) * 2 - 1) * scale
