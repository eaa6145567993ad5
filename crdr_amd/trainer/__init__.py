"""build_trainer: `trainer.type` -> TRAINER_REGISTRY, remaining keys -> kwargs (src/trainer/__init__.py:10-28)."""
import os.path as osp
from copy import deepcopy

from crdr_amd.utils.misc import import_modules
from crdr_amd.utils.registry import TRAINER_REGISTRY

import_modules("crdr_amd.trainer", osp.dirname(osp.abspath(__file__)), suffix="_trainer.py")


def build_trainer(opt):
    if opt.get("trainer"):
        t = deepcopy(opt["trainer"])
        t = t.to_dict() if hasattr(t, "to_dict") else dict(t)
        cls = TRAINER_REGISTRY.get(t.pop("type"))
        return cls(opt, **t)
    raise ValueError('"trainer_type" key is not supported. Please use trainer.type')
