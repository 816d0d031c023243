"""Tracing / metrics (the reference has none: `print` only).

* `range(name)`   : rocTX range (shows up in `rocprofv3 --marker-trace`) + optional wall-clock log;
* `StepMeter`     : images/s over a sliding window without forcing a device sync every step;
* `JsonlLogger`   : one JSON object per line (loss curves, accuracies, throughput) for later plots;
* `rocprof_command`: the profiling recipe used for profiles/*.md.
"""
from __future__ import annotations

import contextlib
import json
import os
import time
from collections import deque
from typing import Optional

import torch

_ENABLED = os.environ.get("DDL_TRACE", "0") == "1"
_LOG = []


@contextlib.contextmanager
def range(name: str):  # noqa: A001  (mirrors torch.cuda.nvtx.range)
    """Named range: a rocTX push/pop on device runs (torch.cuda.nvtx maps to roctx on ROCm) and,
    with DDL_TRACE=1, a host wall-clock record retrievable with `records()`."""
    use_tx = torch.cuda.is_available()
    if use_tx:
        torch.cuda.nvtx.range_push(name)
    t0 = time.perf_counter()
    try:
        yield
    finally:
        if use_tx:
            torch.cuda.nvtx.range_pop()
        if _ENABLED:
            _LOG.append((name, t0, time.perf_counter() - t0))


def records():
    return list(_LOG)


class StepMeter:
    """`with meter: step()`; `images_per_sec()` averages the last `window` steps.  On device the
    timing uses events recorded on the current stream, read lazily (no per-step synchronize)."""

    def __init__(self, images_per_step: int, window: int = 50):
        self.n = images_per_step
        self.samples = deque(maxlen=window)
        self._gpu = torch.cuda.is_available()
        self._pending = deque()
        self.steps = 0

    def __enter__(self):
        if self._gpu:
            self._e0 = torch.cuda.Event(enable_timing=True)
            self._e0.record()
        else:
            self._t0 = time.perf_counter()
        return self

    def __exit__(self, *a):
        self.steps += 1
        if self._gpu:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._pending.append((self._e0, e1))
            while self._pending and self._pending[0][1].query():
                s, e = self._pending.popleft()
                self.samples.append(s.elapsed_time(e) * 1e-3)
        else:
            self.samples.append(time.perf_counter() - self._t0)
        return False

    def images_per_sec(self) -> float:
        if self._gpu and not self.samples and self._pending:
            self._pending[-1][1].synchronize()
            while self._pending:
                s, e = self._pending.popleft()
                self.samples.append(s.elapsed_time(e) * 1e-3)
        if not self.samples:
            return 0.0
        return self.n * len(self.samples) / sum(self.samples)


class JsonlLogger:
    def __init__(self, path: Optional[str]):
        self.f = open(path, "a") if path else None

    @property
    def enabled(self) -> bool:
        return self.f is not None

    def log(self, **kv):
        if self.f:
            kv.setdefault("ts", time.time())
            self.f.write(json.dumps(kv) + "\n")
            self.f.flush()

    def close(self):
        if self.f:
            self.f.close()
            self.f = None


def rocprof_command(cmd: str, out_dir: str = "gpurun_out/prof", pmc: Optional[str] = None) -> str:
    """Shell line for a kernel-trace (or single-pass PMC) profile of `cmd` (never both at once)."""
    if pmc:
        return f"cd /tmp && TMPDIR=/tmp rocprofv3 --pmc {pmc} -d {out_dir} -- {cmd}"
    return f"cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d {out_dir} -- {cmd}"
