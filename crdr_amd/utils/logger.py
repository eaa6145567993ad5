"""Small logging helpers with the reference's call surface (src/utils/logger.py:16-205): a root logger with an
optional file sink, indented sections, key/value dumps, AvgMeter and CSVLogger."""
from __future__ import annotations

import csv
import logging
import os
from contextlib import ContextDecorator
from typing import Dict, Optional

_LOGGER_NAME = "crdr"
_indent = 0
_configured = False


def get_root_logger(log_level: str = "INFO", log_file: Optional[str] = None) -> logging.Logger:
    global _configured
    logger = logging.getLogger(_LOGGER_NAME)
    if not _configured:
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s", "%H:%M:%S"))
        logger.addHandler(handler)
        logger.propagate = False
        logger.setLevel(os.environ.get("CRDR_LOG_LEVEL", "WARNING"))
        _configured = True
    if log_file is not None:
        logger.setLevel(getattr(logging, log_level))
        fh = logging.FileHandler(log_file)
        fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s"))
        logger.addHandler(fh)
    return logger


def _pad() -> str:
    return "  " * _indent


def bolded_log(msg: str, level: str = "INFO", new_line: bool = False, prefix: str = "===== ", suffix: str = " =====") -> None:
    text = f"{prefix}{msg}{suffix}"
    get_root_logger().log(getattr(logging, level), ("\n" if new_line else "") + text)


def log_dict_items(dic: Dict, level: str = "INFO", indent: bool = False) -> None:
    lvl = getattr(logging, level)
    pad = _pad() + ("  " if indent else "")
    for k, v in dic.items():
        get_root_logger().log(lvl, f"{pad}{k}: {v}")


class IndentedLog(ContextDecorator):
    def __init__(self, level: str = "INFO", msg: Optional[str] = None):
        self.level, self.msg = level, msg

    def __enter__(self):
        global _indent
        if self.msg:
            get_root_logger().log(getattr(logging, self.level), _pad() + self.msg)
        _indent += 1
        return self

    def __exit__(self, *exc):
        global _indent
        _indent -= 1
        return False


class AvgMeter:
    """Running mean of named scalars between two `reset()` calls."""

    def __init__(self):
        self.reset()

    def reset(self) -> None:
        self._sum: Dict[str, float] = {}
        self._cnt: Dict[str, int] = {}

    def update(self, values: Dict[str, float]) -> None:
        for k, v in values.items():
            self._sum[k] = self._sum.get(k, 0.0) + float(v)
            self._cnt[k] = self._cnt.get(k, 0) + 1

    def get_avg_values(self) -> Dict[str, float]:
        return {k: self._sum[k] / self._cnt[k] for k in self._sum}


class CSVLogger:
    """Appends dict rows to a CSV file; the header is the key set of the first row."""

    def __init__(self, log_path: str, resume: bool = False):
        self.log_path = log_path
        self._header = None
        if resume and os.path.exists(log_path):
            with open(log_path) as f:
                first = f.readline().strip()
            self._header = first.split(",") if first else None
        elif os.path.exists(log_path):
            os.remove(log_path)

    def update(self, row: Dict) -> None:
        new = self._header is None
        if new:
            self._header = list(row.keys())
        with open(self.log_path, "a", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=self._header, extrasaction="ignore")
            if new:
                wr.writeheader()
            wr.writerow(row)
