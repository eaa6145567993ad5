"""Wall-clock progress statistics (src/utils/timer.py:4-44)."""
from __future__ import annotations

import datetime as _dt
from typing import Dict


def _dhm(td: _dt.timedelta) -> str:
    minutes = td.seconds // 60
    return f"{td.days}d {minutes // 60}h {minutes % 60}m"


def _mdhm(t: _dt.datetime) -> str:
    return f"{t.month}/{t.day:02d} {t.hour:02d}:{t.minute:02d}"


class Timer:
    def __init__(self, start_iter: int, end_iter: int):
        self.start_iter, self.total_iter = start_iter, end_iter - start_iter
        self._t0 = self._last = None

    def start(self) -> None:
        self._t0 = self._last = _dt.datetime.now()

    def get_time_stat(self, current_iter: int) -> Dict[str, str]:
        assert self._t0 is not None, "Timer has not been started"
        now = _dt.datetime.now()
        done = current_iter - self.start_iter
        runtime, interval = now - self._t0, now - self._last
        per_iter = runtime / max(done, 1)
        remaining = per_iter * (self.total_iter - done)
        self._last = now
        return dict(start_time=_mdhm(self._t0), runtime=_dhm(runtime), interval=_dhm(interval),
                    time_per_iter=_dhm(per_iter), remaining=_dhm(remaining), end_time=_mdhm(now + remaining))
