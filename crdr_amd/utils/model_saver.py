"""Rolling checkpoints: torch.save({'iter': i, key: state_dict}) per label, delete the previous one unless its
iteration is in keep_step (src/utils/model_saver.py:9-63)."""
from __future__ import annotations

import os
from typing import Dict, Iterable, Union

import torch

from .logger import get_root_logger
from .path import PathHandler


class Saver:
    def __init__(self, ckpt_root: str, exp: str, save_step: int, keep_step: Union[int, Iterable[int]]):
        self.paths = PathHandler(ckpt_root, exp)
        self.save_step = save_step
        self.keep_step = keep_step if isinstance(keep_step, int) else set(keep_step)

    def _should_keep(self, itr: int) -> bool:
        if isinstance(self.keep_step, int):
            return itr % self.keep_step == 0
        return itr in self.keep_step

    def save(self, network_dict: Dict, save_label: str, current_iter: int, keep: bool) -> None:
        payload = {"iter": current_iter}
        payload.update({k: net.state_dict() for k, net in network_dict.items()})
        torch.save(payload, self.paths.get_ckpt_path(save_label, current_iter))
        prev = current_iter - self.save_step
        if prev == 0:
            return
        if not keep or not self._should_keep(prev):
            old = self.paths.get_ckpt_path(save_label, prev)
            if os.path.exists(old):
                os.remove(old)
            else:
                get_root_logger().warning(f'checkpoint "{old}" to delete does not exist')
