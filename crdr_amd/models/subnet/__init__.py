"""build_subnet: `type` key -> registry lookup, remaining keys -> constructor kwargs (src/models/subnet/__init__.py:16-43)."""
from copy import deepcopy
from typing import Dict

from crdr_amd.utils.registry import (CONTEXTMODEL_REGISTRY, DECODER_REGISTRY, ENCODER_REGISTRY, ENTROPYMODEL_REGISTRY,
                                     HYPERDECODER_REGISTRY, HYPERENCODER_REGISTRY)

from . import autoencoder, context_model, entropy_model, hyperprior  # noqa: F401  (registration side effects)

_REGISTRIES = {
    "encoder": ENCODER_REGISTRY,
    "decoder": DECODER_REGISTRY,
    "hyperencoder": HYPERENCODER_REGISTRY,
    "hyperdecoder": HYPERDECODER_REGISTRY,
    "context_model": CONTEXTMODEL_REGISTRY,
    "entropy_model": ENTROPYMODEL_REGISTRY,
}


def build_subnet(subnet_opt: Dict, subnet_type: str):
    opt = deepcopy(subnet_opt)
    opt = opt.to_dict() if hasattr(opt, "to_dict") else dict(opt)
    cls = _REGISTRIES[subnet_type].get(opt.pop("type"))
    return cls(**opt)
