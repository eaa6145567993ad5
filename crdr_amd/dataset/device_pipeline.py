"""Device-side training input pipeline (reference: data_transform.py:19-45 + base_trainer.py:74-80).

The reference decodes a JPEG, pads / crops / flips it with torchvision and normalises it, per sample, in 8 DataLoader
workers.  On an MI355X the decoded images of a training set fit in HBM (288 GB: ~370 k RGB images of 512x512), so the
pool is decoded ONCE into a flat uint8 buffer on the device and every batch is cut by one launch
(`crdr_crop_flip_normalize`) straight into NHWC fp32 -- no host pixels move per step; only the 6 integers per sample that
describe the random draw do.  The draws follow torchvision's semantics: `RandomCrop(size, pad_if_needed=True,
padding_mode='reflect')` pads BOTH sides by (size - dim) when a side is too short, the crop origin is uniform over the
padded image, and the flip has probability 0.5."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from crdr_amd.hip import lib as L
from crdr_amd.hip import ops


def draw_crops(shapes: Sequence[Sequence[int]], idx: np.ndarray, size: int, rng: np.random.Generator) -> np.ndarray:
    """-> int64 [len(idx)][5] = (H, W, sy0, sx0, flip) for the images `idx` (crop origin in un-padded coordinates)."""
    out = np.zeros((len(idx), 5), dtype=np.int64)
    for k, i in enumerate(idx):
        h, w = shapes[i]
        ph, pw = max(0, size - h), max(0, size - w)            # padding added on BOTH sides (torchvision F.pad [p, p])
        assert ph < h and pw < w, "reflect padding needs pad < image side"
        y0 = int(rng.integers(0, h + 2 * ph - size + 1))
        x0 = int(rng.integers(0, w + 2 * pw - size + 1))
        out[k] = (h, w, y0 - ph, x0 - pw, int(rng.random() < 0.5))
    return out


class DeviceImagePool:
    """Decoded uint8 RGB images, back to back in one device buffer."""

    def __init__(self, images: List[np.ndarray], device):
        assert len(images) > 0
        self.shapes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
        sizes = [h * w * 3 for h, w in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(im, dtype=np.uint8).reshape(-1) for im in images])
        self.pool = torch.from_numpy(flat).to(device)
        self.device = self.pool.device

    @classmethod
    def from_paths(cls, paths: Sequence[str], device) -> "DeviceImagePool":
        from PIL import Image
        return cls([np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8) for p in paths], device)

    def __len__(self):
        return len(self.shapes)


class DeviceCropLoader:
    """Endless iterator of training batches cut on the device: {"real_images": [N, 3, size, size] (NHWC memory)}.

    Every rank of a data-parallel job passes its own `seed` (or the same one plus rank-strided `idx` draws via `rank` /
    `world_size`) so that ranks see different crops."""

    def __init__(self, pool: DeviceImagePool, batch_size: int, size: int = 256, seed: int = 0, rank: int = 0, world_size: int = 1):
        self.pool, self.bs, self.size = pool, batch_size, size
        self.rng = np.random.default_rng([seed, rank])
        self.perm, self.cursor = None, 0
        self.items = torch.zeros((batch_size, 6), dtype=torch.int64, device=pool.device)

    def _next_indices(self) -> np.ndarray:
        out = []
        while len(out) < self.bs:  # shuffle=True, drop_last=True (base_trainer.py:74-80)
            if self.perm is None or self.cursor >= len(self.perm):
                self.perm, self.cursor = self.rng.permutation(len(self.pool)), 0
            out.append(int(self.perm[self.cursor]))
            self.cursor += 1
        return np.asarray(out)

    def cut(self, idx: np.ndarray, draws: np.ndarray, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One launch: images `idx` with draws [N][5] = (H, W, sy0, sx0, flip) -> [N, 3, size, size]."""
        lib = L.load()
        n = len(idx)
        table = np.concatenate([self.pool.offsets[idx][:, None], draws], axis=1).astype(np.int64)
        items = self.items[:n]
        items.copy_(torch.from_numpy(table))
        if out is None:
            out = ops.empty_nhwc(n, 3, self.size, self.size, self.pool.device)
        L.check(lib.crdr_crop_flip_normalize(self.pool.pool.data_ptr(), items.data_ptr(), n, self.size, self.size, out.data_ptr(),
                                             ops.ld_for(3), ops._stream()), "crop_flip_normalize")
        return out

    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, torch.Tensor]:
        idx = self._next_indices()
        return {"real_images": self.cut(idx, draw_crops(self.pool.shapes, idx, self.size, self.rng))}

    def __len__(self):
        return 1 << 30
