"""Name -> class registries: the drop-in plug-in boundary of the codec.

Same surface as the reference (src/utils/registry.py:12-95): `@X_REGISTRY.register()` stores the decorated
class under its `__name__`; `X_REGISTRY.get(name)` returns it or raises KeyError; duplicate names are an error.
"""
from __future__ import annotations

import inspect
import os
from typing import Any, Callable, Dict, List, Optional

from .logger import get_root_logger


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._entries: Dict[str, Dict[str, Any]] = {}

    @property
    def name(self) -> str:
        return self._name

    def register(self) -> Callable[[Any], Any]:
        """Decorator (or plain call: `REG.register()(cls)`) registering `obj` under `obj.__name__`."""
        caller_file = os.path.basename(inspect.stack()[1].filename)

        def _add(obj: Any) -> Any:
            key = obj.__name__
            if key in self._entries:
                raise AssertionError(f"An object named '{key}' was already registered in '{self._name}' registry!")
            self._entries[key] = {"obj": obj, "filename": caller_file}
            return obj

        return _add

    def get(self, class_name: str, display_name: Optional[str] = None) -> Any:
        entry = self._entries.get(class_name)
        if entry is None:
            raise KeyError(f"No object named '{class_name}' found in '{self._name}' registry!")
        shown = self._name if display_name is None else display_name
        get_root_logger().info(f"{shown} [{class_name}] (from {entry['filename']}) is built")
        return entry["obj"]

    def __contains__(self, name: str) -> bool:
        return name in self._entries

    def keys(self) -> List[str]:
        return list(self._entries)


TRAINER_REGISTRY = Registry("trainer")
OPTIMIZER_REGISTRY = Registry("optimizer")
SCHEDULER_REGISTRY = Registry("scheduler")

MODEL_REGISTRY = Registry("comp_model")
ENCODER_REGISTRY = Registry("encoder")
DECODER_REGISTRY = Registry("decoder")
HYPERENCODER_REGISTRY = Registry("hyperencoder")
HYPERDECODER_REGISTRY = Registry("hyperdecoder")
CONTEXTMODEL_REGISTRY = Registry("context_model")
ENTROPYMODEL_REGISTRY = Registry("entropy_model")
DISCRIMINATOR_REGISTRY = Registry("discriminator")

DATASET_REGISTRY = Registry("dataset")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")
