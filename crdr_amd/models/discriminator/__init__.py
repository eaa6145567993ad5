"""build_discriminator (src/models/discriminator/__init__.py:15-28)."""
import os.path as osp
from copy import deepcopy
from typing import Dict

from crdr_amd.utils.misc import import_modules
from crdr_amd.utils.registry import DISCRIMINATOR_REGISTRY

import_modules("crdr_amd.models.discriminator", osp.dirname(osp.abspath(__file__)), suffix="_discriminator.py")


def build_discriminator(discriminator_opt: Dict):
    opt = deepcopy(discriminator_opt)
    opt = opt.to_dict() if hasattr(opt, "to_dict") else dict(opt)
    return DISCRIMINATOR_REGISTRY.get(opt.pop("type"))(**opt)
