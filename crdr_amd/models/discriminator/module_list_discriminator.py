"""One sub-discriminator per rate level, selected by int(rate_ind)
(src/models/discriminator/module_list_discriminator.py:14-30)."""
from __future__ import annotations

from typing import Union

import torch
import torch.nn as nn
from torch import Tensor

from crdr_amd.utils.registry import DISCRIMINATOR_REGISTRY

from .base_discriminator import BaseDiscriminator


@DISCRIMINATOR_REGISTRY.register()
class ModuleListDiscriminator(BaseDiscriminator):
    def __init__(self, _subd_type, _num_subd, **kwargs) -> None:
        super().__init__()
        self.subD_list = nn.ModuleList(DISCRIMINATOR_REGISTRY.get(_subd_type)(**kwargs) for _ in range(_num_subd))

    def forward(self, input, rate_ind: Union[float, Tensor], **kwargs):
        if isinstance(rate_ind, torch.Tensor):
            assert rate_ind.numel() == 1
            rate_ind = rate_ind.item()
        return self.subD_list[int(rate_ind)](input, **kwargs)
