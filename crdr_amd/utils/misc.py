"""import_modules (auto-registration of *_model.py etc., src/utils/misc.py:38-45) and dict pretty printing."""
from __future__ import annotations

import importlib
import os.path as osp
from glob import glob


def import_modules(base_pkg: str, search_dir: str, suffix: str = ".py") -> None:
    for path in sorted(glob(osp.join(search_dir, f"*{suffix}"))):
        importlib.import_module(f"{base_pkg}.{osp.splitext(osp.basename(path))[0]}")


def dict2str(dic, level: int = 0, indent_width: int = 2) -> str:
    lines = []
    for k, v in dic.items():
        pad = " " * (indent_width * (level + 1))
        if isinstance(v, dict):
            lines.append(f"{pad}{k}:")
            lines.append(dict2str(v, level + 1, indent_width))
        elif isinstance(v, (list, tuple)):
            lines.append(f"{pad}{k}:")
            lines.extend(f"{pad}  - {item}" for item in v)
        else:
            lines.append(f"{pad}{k}: {v}")
    return "\n".join(lines)
