"""Input pipeline. The hot path is benchmarked on device-resident synthetic crops; the
folder datasets reproduce the reference's tensor contract -- {'real_images': float32 [N,3,256,256] in [-1,1]}
from RandomCrop(256, reflect pad if needed) + HFlip(0.5) + ToTensor + Normalize(0.5, 0.5)
(src/dataset/data_transform.py:19-45, base_dataset.py:30-34) -- without torchvision."""
from __future__ import annotations

import os
from glob import glob
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from crdr_amd.utils.registry import DATASET_REGISTRY


class SyntheticLoader:
    """Endless iterator of one device-resident batch of uniform [-1,1] noise images (throughput is data independent
    for dense convolutions); a fresh view each step, no host work."""

    def __init__(self, batch_size: int, size: int, device, seed: int = 0, n_batches: int = 4):
        g = torch.Generator().manual_seed(seed)
        self.batches = [(torch.rand(batch_size, 3, size, size, generator=g) * 2 - 1).to(device) for _ in range(n_batches)]
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, torch.Tensor]:
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return {"real_images": b}

    def __len__(self):
        return 1 << 30


def _load_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def _to_tensor(img: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float() / 255.0
    return (t - 0.5) / 0.5


class _FolderDataset(Dataset):
    exts = ("*.png", "*.jpg", "*.jpeg", "*.JPEG")

    def __init__(self, root_dir: str, is_train: bool, image_size: int = 256, subset_list=None, **_):
        dirs = [os.path.join(root_dir, str(s)) for s in subset_list] if (is_train and subset_list and
                                                                         os.path.isdir(os.path.join(root_dir, str(subset_list[0])))) else [root_dir]
        self.paths = sorted(p for d in dirs for e in self.exts for p in glob(os.path.join(d, e)))
        self.is_train, self.size = is_train, image_size

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i: int) -> Dict[str, torch.Tensor]:
        img = _load_rgb(self.paths[i])
        if self.is_train:
            s = self.size
            ph, pw = max(0, s - img.shape[0]), max(0, s - img.shape[1])
            if ph or pw:
                img = np.pad(img, ((0, ph), (0, pw), (0, 0)), mode="reflect")
            y0 = np.random.randint(0, img.shape[0] - s + 1)
            x0 = np.random.randint(0, img.shape[1] - s + 1)
            img = img[y0:y0 + s, x0:x0 + s]
            if np.random.rand() < 0.5:
                img = img[:, ::-1]
        return {"real_images": _to_tensor(img)}


@DATASET_REGISTRY.register()
class OpenImageImageDataset(_FolderDataset):
    pass


@DATASET_REGISTRY.register()
class KodakImageDataset(_FolderDataset):
    pass


_NAMES = {"kodak": "Kodak", "openimage": "OpenImage"}


def build_dataset(dataset_opt: Dict, is_train: bool = True) -> Dataset:
    o = dict(dataset_opt)
    key = _NAMES[o.pop("name").lower()] + o.pop("type")
    ds = DATASET_REGISTRY.get(key)(is_train=is_train, **o)
    assert len(ds) > 0, "len(dataset) should be >0."
    return ds


def build_loaders(opt, device) -> Tuple[object, Optional[DataLoader]]:
    ds = opt.get("dataset", None)
    bs = ds.get("batch_size", 8) if ds else 8
    tr = ds.get("train_dataset", None) if ds else None
    if tr is None or tr.get("type") == "SyntheticDataset" or not os.path.isdir(str(tr.get("root_dir", ""))):
        size = tr.get("image_size", 256) if tr else 256
        return SyntheticLoader(bs, size, device, seed=opt.get("data_seed", 0)), None
    if tr.get("device_pool", False) and str(device).startswith("cuda"):
        # decode the training set once into HBM and cut every batch with one launch (crdr_amd/dataset/device_pipeline.py)
        from .device_pipeline import DeviceCropLoader, DeviceImagePool
        from crdr_amd.trainer import dist as _D
        o = {k: v for k, v in dict(tr).items() if k != "device_pool"}
        paths = build_dataset(o, True).paths
        train = DeviceCropLoader(DeviceImagePool.from_paths(paths, device), bs, int(o.get("image_size", 256)),
                                 seed=int(opt.get("data_seed", 0)), rank=_D.rank(), world_size=_D.world_size())
    else:
        from crdr_amd.trainer import dist as _D
        dset = build_dataset(tr, True)
        gen = torch.Generator().manual_seed(int(opt.get("data_seed", 0)))
        if _D.world_size() > 1:  # disjoint, reproducible shards per rank (same permutation everywhere, strided by rank)
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dset, num_replicas=_D.world_size(), rank=_D.rank(), shuffle=True, seed=int(opt.get("data_seed", 0)),
                                         drop_last=True)
            train = DataLoader(dset, batch_size=bs, drop_last=True, sampler=sampler, num_workers=opt.get("num_workers", 8))
        else:
            train = DataLoader(dset, batch_size=bs, drop_last=True, shuffle=True, generator=gen, num_workers=opt.get("num_workers", 8))
    ev = ds.get("eval_dataset", None)
    evl = DataLoader(build_dataset(ev, False), batch_size=1, shuffle=False, num_workers=1) if ev and os.path.isdir(str(ev.get("root_dir", ""))) else None
    return train, evl
