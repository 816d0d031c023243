"""hipGraph capture helpers (torch.cuda.CUDAGraph is a hipGraph on ROCm).

Capturing next to a live RCCL process group needs two precautions:
  * torch's collective watchdog thread polls events of outstanding collectives; an event query from
    another thread invalidates a capture made in the default "global" error mode, so captures use
    "thread_local" whenever torch.distributed is up;
  * no collective may still be in flight when a capture starts (`quiesce`).
"""
from __future__ import annotations

import time

import torch


def _dist_up() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def quiesce():
    """Drain the device and give the collective watchdog a poll interval to retire finished work."""
    torch.cuda.synchronize()
    if _dist_up():
        time.sleep(0.25)
        torch.cuda.synchronize()


class CaptureError(RuntimeError):
    """A stream capture failed for a reason of the runtime's (an operation that is not permitted while capturing,
    a communicator that cannot be captured, an invalidated capture).  The only kind of set-up failure a caller may
    answer by stepping down to a less-captured mode; programming errors inside the captured function (NameError,
    TypeError, ...) are never wrapped and propagate as they are."""


def capture(fn, pool=None):
    """Record `fn()` into a new graph; returns (graph, fn's return value)."""
    g = torch.cuda.CUDAGraph()
    mode = "thread_local" if _dist_up() else "global"
    try:
        with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
            out = fn()
    except RuntimeError as e:           # torch.AcceleratorError / DistBackendError are RuntimeErrors too
        if isinstance(e, (NotImplementedError, RecursionError)):
            raise
        raise CaptureError(f"hipGraph capture failed: {e}") from e
    return g, out
