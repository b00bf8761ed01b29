"""Shared plumbing of the op wrappers."""
from __future__ import annotations

import torch

from .. import _native


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t: torch.Tensor) -> int:
    """Handle of the current HIP stream of ``t``'s device.  The raw query is used when the build has it:
    constructing a ``torch.cuda.Stream`` object costs 4–5 µs, and a generation asks a dozen times."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device).cuda_stream


def on_gpu(*tensors) -> bool:
    """True when the op must run its HIP kernel.  Mixed placement is an error."""
    flags = [t.is_cuda for t in tensors if isinstance(t, torch.Tensor)]
    if not flags:
        return False
    if any(flags) and not all(flags):
        raise ValueError("op received a mix of host and device tensors")
    return flags[0]


def hip():
    return _native.hip()


def check(status, what):
    _native.check(status, what)


DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}
