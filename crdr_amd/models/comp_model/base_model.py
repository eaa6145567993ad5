"""Shared model plumbing: image pre/post-processing, checkpoint loading with the reference's key schema,
aux-parameter split (src/models/comp_model/base_model.py:16-170)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from crdr_amd.hip import ops
from crdr_amd.models.subnet.entropy_model.entropy_bottleneck import EntropyBottleneck
from crdr_amd.models.subnet.entropy_model.gaussian_conditional import GaussianMeanScaleConditional
from crdr_amd.utils.logger import get_root_logger

_EB_RENAMES = {"matrices": "_matrix", "biases": "_bias", "factors": "_factor"}  # compressai >= 1.2 ParameterList names


def _resize_registered_buffers(module: nn.Module, prefix: str, names, state_dict) -> None:
    """Buffers whose size depends on the trained model (CDF tables) take the checkpoint's shape before loading
    (what compressai.models.utils.update_registered_buffers does for base_model.py:80-96)."""
    for n in names:
        key = f"{prefix}.{n}"
        if key in state_dict:
            cur = getattr(module, n)
            new = state_dict[key]
            if cur.shape != new.shape:
                setattr(module, n, torch.empty(new.shape, dtype=new.dtype if cur.numel() == 0 else cur.dtype, device=cur.device))


class BaseModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.device = opt.device
        self.convert_img_range = opt.get("convert_img_range_to_01", False)
        self._build_subnets()
        self.stride = 64

    def _build_subnets(self) -> None:
        raise NotImplementedError()

    # ---- images
    def data_preprocess(self, real_images: Tensor, is_train: bool = True) -> Tensor:
        out = real_images
        if self.convert_img_range:
            out = (out + 1.0) / 2.0
        if not is_train:
            out = self.pad_images(out)
        out = out.to(self.device)
        out, _ = ops.nhwc(out)  # NHWC memory, channels padded to 4 (zero lane)
        return out

    def data_postprocess(self, *images: Tensor, size: Tuple[int, int], is_train: bool):
        H, W = size
        out = []
        for img in images:
            if self.convert_img_range:
                img = (img - 0.5) * 2.0
            if not is_train:
                img = self._crop_image(img, H, W).clamp(-1, 1)
            out.append(img)
        return out[0] if len(out) == 1 else tuple(out)

    def pad_images(self, *images: Tensor):
        out = [self._pad_image(img, self.stride, mode="reflect") for img in images]
        return out[0] if len(out) == 1 else tuple(out)

    @staticmethod
    def _pad_image(x: Tensor, stride: int, mode: str = "reflect") -> Tensor:
        _, _, H, W = x.size()
        padW = int(np.ceil(W / stride) * stride - W)
        padH = int(np.ceil(H / stride) * stride - H)
        if padH == 0 and padW == 0:
            return x
        return F.pad(x, (0, padW, 0, padH), mode=mode)

    def crop_images(self, *images: Tensor, size: Tuple[int, int]):
        H, W = size
        out = [self._crop_image(img, H, W) for img in images]
        return out[0] if len(out) == 1 else tuple(out)

    @staticmethod
    def _crop_image(x: Tensor, H: int, W: int) -> Tensor:
        return x[:, :, :H, :W]

    # ---- parameters
    def aux_loss(self) -> Tensor:
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def separate_aux_parameters(self) -> Tuple[Dict, Dict]:
        named = {n: p for n, p in self.named_parameters() if p.requires_grad}
        main = {n: named[n] for n in sorted(named) if not n.endswith(".quantiles")}
        aux = {n: named[n] for n in sorted(named) if n.endswith(".quantiles")}
        assert len(main) + len(aux) == len(named)
        return main, aux

    @staticmethod
    def _canonical_keys(state_dict) -> "OrderedDict":
        """compressai >= 1.2.5 stores the factorised prior's parameters as ParameterLists (`matrices.0`); 1.2.4 -- the
        reference's pin -- and this package use `_matrix0`."""
        sd = OrderedDict()
        for k, v in state_dict.items():
            parts = k.split(".")
            if len(parts) >= 3 and parts[-2] in _EB_RENAMES and parts[-1].isdigit():  # entropy_model_z.matrices.0
                k = ".".join(parts[:-2] + [f"{_EB_RENAMES[parts[-2]]}{parts[-1]}"])
            sd[k] = v
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = self._canonical_keys(state_dict)
        if isinstance(getattr(self, "entropy_model_z", None), EntropyBottleneck):
            _resize_registered_buffers(self.entropy_model_z, "entropy_model_z", ["_quantized_cdf", "_offset", "_cdf_length"], sd)
        if isinstance(getattr(self, "entropy_model_y", None), GaussianMeanScaleConditional):
            _resize_registered_buffers(self.entropy_model_y, "entropy_model_y",
                                       ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], sd)
        return super().load_state_dict(sd, strict=strict)

    def load_learned_weight(self, ckpt_path: str) -> None:
        get_root_logger().info(f"load checkpoint: {ckpt_path}")
        ckpt = torch.load(ckpt_path, map_location="cpu")
        incoming = self._canonical_keys(OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in ckpt["comp_model"].items()))
        own = self.state_dict()
        merged = dict(own)
        merged.update({k: v for k, v in incoming.items() if k in own})  # key intersection (warm start across stages)
        self.load_state_dict(merged)
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                m.update(force=False)

    def codec_setup(self):
        raise NotImplementedError()
