"""Lightweight tracing helpers.

``scope(name)`` marks a region with a roctx range (``torch.cuda.nvtx`` is backed by roctx on
ROCm), so that ``rocprofv3 --marker-trace`` attributes kernels to ask / evaluate / tell.  Ranges
are only emitted when ``EVOX_AMD_PROFILE=1`` (or ``enable()`` was called): the default path has
no overhead.  ``StepTimer`` measures device time of a callable with events (no host stalls inside
the measured region)."""
from __future__ import annotations

import contextlib
import os

import torch

_enabled = os.environ.get("EVOX_AMD_PROFILE", "0") == "1"


def enable(flag: bool = True):
    global _enabled
    _enabled = flag


def enabled() -> bool:
    return _enabled


@contextlib.contextmanager
def scope(name: str):
    if _enabled and torch.cuda.is_available():
        torch.cuda.nvtx.range_push(name)
        try:
            yield
        finally:
            torch.cuda.nvtx.range_pop()
    else:
        yield


class StepTimer:
    """Average device milliseconds of ``fn()`` over ``iters`` calls (events on the current stream)."""

    def __init__(self, warmup: int = 3, iters: int = 10):
        self.warmup, self.iters = warmup, iters

    def __call__(self, fn):
        for _ in range(self.warmup):
            fn()
        if not torch.cuda.is_available():
            import time
            t0 = time.perf_counter()
            for _ in range(self.iters):
                fn()
            return (time.perf_counter() - t0) / self.iters * 1e3
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(self.iters):
            fn()
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) / self.iters
