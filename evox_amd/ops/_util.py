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


def pick(x: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
    """``x[i]`` for a 0-d index tensor without the host read that tensor indexing does for it (so the expression can be
    recorded into a HIP graph)."""
    return x.index_select(0, i.reshape(1))[0]


def count_groups(group: torch.Tensor, n_groups: int) -> torch.Tensor:
    """``torch.bincount(group, minlength=n_groups)`` for indices known to lie in ``[0, n_groups)``, without the host
    read by which ``bincount`` sizes its result (so the expression can be recorded into a HIP graph)."""
    out = torch.zeros(n_groups, dtype=torch.int64, device=group.device)
    return out.index_add_(0, group, torch.ones_like(group, dtype=torch.int64))
