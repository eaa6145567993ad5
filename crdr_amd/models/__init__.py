"""build_comp_model(opt): `model_type` -> MODEL_REGISTRY (src/models/__init__.py:21-34)."""
from copy import deepcopy

from crdr_amd.utils.registry import MODEL_REGISTRY

from . import comp_model, subnet  # noqa: F401  (registration side effects)

__all__ = ["build_comp_model", "build_trained_comp_model"]


def build_comp_model(opt):
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt["model_type"])(opt)


def build_trained_comp_model(opt, ckpt_path: str):
    model = build_comp_model(opt)
    model.load_learned_weight(ckpt_path=ckpt_path)
    return model
