"""PSNR / image writing with the reference's uint8 TRUNCATION (not rounding) semantics
(src/utils/img_utils.py:17-42, 79-132): [-1,1] float -> (x+1)/2*255 -> astype(uint8)."""
from __future__ import annotations

import numpy as np
import torch


def tensor2img(t: torch.Tensor) -> np.ndarray:
    """[1,3,H,W] in [-1,1] -> HWC uint8 by truncation."""
    x = t.detach().float().cpu()
    if x.dim() == 4:
        assert x.shape[0] == 1
        x = x[0]
    x = ((x + 1.0) / 2.0 * 255.0).permute(1, 2, 0).numpy()
    return x.astype(np.uint8)


def calc_psnr(real: torch.Tensor, fake: torch.Tensor, max_val: float = 255) -> float:
    """src/utils/img_utils.py:102-132 arithmetic, step for step: both images to [0, 255] by (x + 1) / 2 * 255, TRUNCATED to
    uint8, float32 mean of the squared float32 differences, 10 log10(255^2 / mse) in Python floats (pinned by
    tests/golden/reference_metrics.json, recorded from the reference's own function)."""
    import math
    assert max_val == 255
    a = tensor2img(real).astype(np.float32)
    b = tensor2img(fake).astype(np.float32)
    mse = np.mean(np.power(a - b, 2))
    if mse == 0:
        return float("inf")
    return 10.0 * math.log10((255.0 ** 2) / mse)


def imwrite(path: str, t: torch.Tensor) -> None:
    from PIL import Image
    Image.fromarray(tensor2img(t)).save(path)
