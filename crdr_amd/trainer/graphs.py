"""HIP-graph execution of the training step.

A step is a fixed sequence of a few thousand kernel launches whose shapes depend only on (batch shape, rate index),
so after an eager warm-up (which also fills the autotuner's cache) each *segment* of the step is captured once per
rate index into a hipGraph and replayed; the Python / ctypes launch path -- which otherwise bounds the step at
~170 ms on MI355X -- disappears.  Collectives (RCCL all-reduce of the flat gradient buffers) stay outside the
graphs, between segments, so single- and multi-GPU runs execute the same graphs.

Everything that varies between iterations enters through device tensors that are refreshed before the replay:
the image batch, the beta Fourier features, the beta loss weight, lr / step count / skip flag of the optimisers.
Random noise is drawn inside the graphs with torch's graph-safe Philox generator."""
from __future__ import annotations

from typing import Callable, Dict, Hashable

import torch

from crdr_amd.hip import functional as HF
from crdr_amd.hip import ops


class SegmentGraphs:
    def __init__(self, enabled: bool):
        self.enabled = enabled and torch.cuda.is_available()
        self._graphs: Dict[Hashable, torch.cuda.CUDAGraph] = {}
        self._outs: Dict[Hashable, object] = {}
        self._hooks: Dict[Hashable, list] = {}
        self._pool = None
        self._stream = torch.cuda.Stream() if self.enabled else None

    @property
    def stream(self):
        return self._stream

    def step_scope(self):
        """Context manager for one training iteration: with graphs enabled ALL work of the trainer -- eager warm-up
        and capture alike -- is issued on one dedicated stream, so autograd nodes that outlive an iteration (gradient
        accumulators) never belong to a different stream than the capture."""
        return _StepScope(self._stream) if self.enabled else _Null()

    def run(self, key: Hashable, fn: Callable[[], object]):
        """Run `fn` eagerly (graphs disabled) or through its captured graph (captured on first use)."""
        if not self.enabled:
            return fn()
        g = self._graphs.get(key)
        fresh = False
        if g is None:
            g = torch.cuda.CUDAGraph()
            ops.reserve_workspace(torch.cuda.current_device(), self._stream)
            torch.cuda.synchronize()
            # with a process group alive, its watchdog thread polls events while we capture: only this thread's
            # unsafe calls should abort the capture
            import torch.distributed as dist
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            import os
            dbg = os.environ.get("CRDR_DEBUG_DIST", "0") == "1"
            if dbg:
                print(f"[graphs] capture begin {key} stream={self._stream.cuda_stream:#x}", flush=True)
            ops.REPLAY_HOOKS = hooks = []
            try:
                with torch.cuda.graph(g, pool=self._pool, stream=self._stream, capture_error_mode=mode):
                    out = fn()
            finally:
                ops.REPLAY_HOOKS = None
            self._hooks[key] = hooks
            fresh = True
            if dbg:
                print(f"[graphs] capture end {key}", flush=True)
            self._pool = g.pool()
            self._graphs[key] = g
            self._outs[key] = out
        g.replay()
        if not fresh:   # host-side bookkeeping of the code the graph replaced (ops.on_replay): on the capture pass that code itself ran
            for h in self._hooks.get(key, ()):
                h()
        return self._outs[key]

    def __len__(self):
        return len(self._graphs)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _StepScope:
    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.stream.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        self.ctx.__exit__(*a)
        torch.cuda.current_stream().wait_stream(self.stream)
        return False
