"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) or gloo for
the CPU tests. The reference is single-GPU only (README.md:63); this is the MI355X scale-out of its training step:
full replicas, one flat-buffer all-reduce per optimiser, and rank-consistent control decisions."""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch
import torch.distributed as dist


_FORCE = os.environ.get("CRDR_FORCE_DIST", "0") == "1"  # exercise the collective path with a single rank (tests)


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns the local rank."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return local


_DEBUG = os.environ.get("CRDR_DEBUG_DIST", "0") == "1"


def _dbg(what: str) -> None:
    if _DEBUG:
        cap = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
        print(f"[dist] {what} stream={torch.cuda.current_stream().cuda_stream:#x} capturing={cap}", flush=True)


_comm_streams = {}


@contextlib.contextmanager
def comm_scope(ref: torch.Tensor):
    """Issue the enclosed collectives on a dedicated communication stream, ordered after / before the current one.

    torch runs a blocking RCCL collective on the *current* stream and hands its completion event to the process-group
    watchdog thread, which polls it with hipEventQuery.  The trainer's current stream is the one its HIP graphs are
    captured on, and HIP refuses to query an event whose stream has since started capturing
    (hipErrorCapturedEvent -> the watchdog aborts the process).  Collectives therefore never touch the capture stream:
    they get a stream of their own that is never captured."""
    if not ref.is_cuda:
        yield
        return
    cur = torch.cuda.current_stream(ref.device)
    comm = _comm_streams.get(ref.device)
    if comm is None:
        comm = _comm_streams[ref.device] = torch.cuda.Stream(ref.device)
    comm.wait_stream(cur)
    with torch.cuda.stream(comm):
        yield
    cur.wait_stream(comm)


def all_reduce_(t: torch.Tensor, op=None) -> None:
    if not is_dist():
        return
    _dbg(f"all_reduce {t.numel()}")
    with comm_scope(t):
        dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM)


def _mean_reduce_(b: torch.Tensor, ws: int) -> None:
    """Average across ranks.  RCCL averages inside the collective (ReduceOp.AVG: inputs scaled by 1/ws on the way in, exact
    for the power-of-two world sizes of one node), which saves a separate pass over the 511 MB buffer on the communication
    stream; other backends (gloo in the CPU tests) sum and divide."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(b, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        b.div_(ws)


def all_reduce_mean_(buffers: List[torch.Tensor]) -> None:
    """In-place average of each flat buffer across ranks (one collective per buffer)."""
    if not is_dist() or not buffers:
        return
    ws = dist.get_world_size()
    with comm_scope(buffers[0]):
        for b in buffers:
            _dbg(f"all_reduce {b.numel()}")
            _mean_reduce_(b, ws)


# Overlap trace (bench.py --gpus N, tests): a list switches it on; every AsyncGradSync then appends
# {"label", "bytes", "ready", "start", "done", "waitpoint"} -- timing events for: the bucket's gradients complete on the compute
# stream / the collective starts on the communication stream / it completes / the compute stream reaches the point where it
# needs the result.  max(0, done - waitpoint) is the part of the all-reduce that was NOT hidden behind compute.
TRACE: Optional[list] = None


def trace_summary(trace: list) -> dict:
    """Per-bucket averages over the traced steps (ms)."""
    by = {}
    for r in trace:
        b = by.setdefault(r["label"], {"MB": r["bytes"] / 1e6, "n": 0, "queue_ms": 0.0, "allreduce_ms": 0.0, "exposed_ms": 0.0, "slack_ms": 0.0})
        b["n"] += 1
        b["queue_ms"] += r["ready"].elapsed_time(r["start"])        # waiting for the communication stream (earlier buckets)
        b["allreduce_ms"] += r["start"].elapsed_time(r["done"])
        gap = r["waitpoint"].elapsed_time(r["done"]) if r.get("waitpoint") is not None else 0.0
        b["exposed_ms"] += max(0.0, gap)                           # compute stream idle until the collective finished
        b["slack_ms"] += max(0.0, -gap)                            # collective finished this long before it was needed
    out = []
    for label, b in by.items():
        n = max(b.pop("n"), 1)
        out.append({"bucket": label, "MB": round(b["MB"], 1), **{k: round(v / n, 3) for k, v in b.items() if k != "MB"},
                    "busbw_GBps": round(b["MB"] / 1e3 / max(b["allreduce_ms"] / n * 1e-3, 1e-9), 1)})
    return {"buckets": out, "exposed_ms_per_step": round(sum(r["exposed_ms"] for r in out), 3),
            "note": "HIP events on the compute and communication streams around every gradient all-reduce; exposed = the compute "
                    "stream reached the point where it needs the bucket before the collective had finished"}


class AsyncGradSync:
    """Mean all-reduce of flat gradient buffers (+ MAX all-reduce of flags) issued on the communication stream WITHOUT
    making the caller's stream wait: independent work (the discriminator's forward/backward while the generator's
    510 MB of gradients travel over xGMI) overlaps the collective; wait() orders the caller's stream behind it."""

    def __init__(self, mean_buffers: List[torch.Tensor], max_flags: Optional[List[torch.Tensor]] = None, label: str = ""):
        ts = list(mean_buffers) + list(max_flags or [])
        self._comm = None
        self._rec = None
        if not is_dist() or not ts:
            return
        ws = dist.get_world_size()

        def issue():
            for b in mean_buffers:
                _dbg(f"all_reduce {b.numel()}")
                _mean_reduce_(b, ws)
            for f in max_flags or []:
                _dbg("all_reduce flag")
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
        if ts[0].is_cuda:
            cur = torch.cuda.current_stream(ts[0].device)
            comm = _comm_streams.get(ts[0].device)
            if comm is None:
                comm = _comm_streams[ts[0].device] = torch.cuda.Stream(ts[0].device)
            tr = TRACE is not None
            if tr:
                self._rec = {"label": label, "bytes": sum(b.numel() * b.element_size() for b in mean_buffers),
                             "ready": torch.cuda.Event(enable_timing=True), "start": torch.cuda.Event(enable_timing=True), "waitpoint": None}
                self._rec["ready"].record(cur)
            comm.wait_stream(cur)
            with torch.cuda.stream(comm):
                if tr:
                    self._rec["start"].record(comm)
                issue()
                self._done = torch.cuda.Event(enable_timing=tr)
                self._done.record(comm)  # later collectives queued on the same stream are not waited for
            if tr:
                self._rec["done"] = self._done
                TRACE.append(self._rec)
            self._comm = comm
        else:
            issue()

    def wait(self) -> None:
        if self._comm is not None:
            cur = torch.cuda.current_stream(self._comm.device)
            if self._rec is not None:
                self._rec["waitpoint"] = torch.cuda.Event(enable_timing=True)
                self._rec["waitpoint"].record(cur)
            cur.wait_event(self._done)
            self._comm = None


def all_reduce_scalars_mean(t: torch.Tensor) -> torch.Tensor:
    if not is_dist():
        return t
    t = t.clone()
    all_reduce_(t)
    return t / dist.get_world_size()


def any_rank_true(flag: bool, device) -> bool:
    """Logical OR across ranks (so that every rank takes the same skip-update decision)."""
    if not is_dist():
        return flag
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    all_reduce_(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def broadcast_module_(module: torch.nn.Module, src: int = 0) -> None:
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel() > 0:
            dist.broadcast(t.data, src=src)
