"""build_loss (src/losses/__init__.py:13-27)."""
import os.path as osp
from copy import deepcopy
from typing import Dict, Optional

from crdr_amd.utils.misc import import_modules
from crdr_amd.utils.registry import LOSS_REGISTRY

import_modules("crdr_amd.losses", osp.dirname(osp.abspath(__file__)), suffix="_loss.py")


def build_loss(opt: Dict, loss_name: Optional[str] = None):
    opt = deepcopy(opt)
    opt = opt.to_dict() if hasattr(opt, "to_dict") else dict(opt)
    return LOSS_REGISTRY.get(opt.pop("type"), display_name=loss_name)(**opt)
