"""Process-level distributed primitives.

One process drives one GPU; processes talk through ``torch.distributed`` (backend ``nccl`` is
RCCL over xGMI on ROCm, ``gloo`` on CPU).  These helpers are the single place the rest of the
framework touches the communicator, and every one of them degrades to the identity in a
single-process run, which is how the reference's ``all_gather(axis_name=None)`` behaves
(reference: ``src/evox/core/distributed.py:17-87``).
"""
from __future__ import annotations

import enum
import os
from typing import Optional, Sequence

import torch

POP_AXIS_NAME = "POP"


class ShardingType(enum.Enum):
    """Placement of a state field across the ranks of a job."""

    SHARED_FIRST_DIM = "shard_dim0"
    REPLICATED = "replicated"

    def get_sharding(self, devices=None):
        """Placement descriptor ``(kind, world_size)`` for ``devices`` (default: the job)."""
        n = len(devices) if devices is not None else get_world_size()
        return (self.value, n)

    def shard(self, x: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
        if self is ShardingType.REPLICATED or world_size == 1:
            return x
        n = x.shape[0]
        if n % world_size:
            raise ValueError(f"dim0={n} not divisible by world_size={world_size}")
        s = n // world_size
        return x[rank * s:(rank + 1) * s]


def _dist():
    import torch.distributed as dist

    return dist


def is_dist_initialized() -> bool:
    d = _dist()
    return d.is_available() and d.is_initialized()


def get_process_id() -> int:
    return _dist().get_rank() if is_dist_initialized() else 0


get_rank = get_process_id


def get_world_size() -> int:
    return _dist().get_world_size() if is_dist_initialized() else 1


def init_process_group(backend: Optional[str] = None, device: Optional[torch.device] = None, **kw):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.

    Picks ``nccl`` (RCCL) when a GPU is visible, else ``gloo``; binds this process to
    ``cuda:LOCAL_RANK``.  Safe to call when already initialised.  The rendezvous and every collective get a finite
    time limit (``timeout=`` as a ``datetime.timedelta`` or seconds; default ``EVOX_AMD_DIST_TIMEOUT`` or 300 s), so
    a peer that never arrives ends in an error instead of a hang.
    """
    import datetime

    d = _dist()
    if d.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    timeout = kw.pop("timeout", None)
    if timeout is None:
        timeout = float(os.environ.get("EVOX_AMD_DIST_TIMEOUT", "300"))
    if not isinstance(timeout, datetime.timedelta):
        timeout = datetime.timedelta(seconds=float(timeout))
    kw["timeout"] = timeout
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local if device is None else device)
        kw.setdefault("device_id", torch.device("cuda", torch.cuda.current_device()))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    d.init_process_group(backend=backend, **kw)


def _p2p_group(x: torch.Tensor):
    """The one-shot peer-to-peer group when it is switched on (``EVOX_AMD_P2P=1``) and ``x`` is a
    device tensor of 4-byte elements that fits its buffers; None otherwise (RCCL / gloo path)."""
    if not x.is_cuda or x.element_size() != 4 or x.numel() == 0:
        return None
    from ..parallel import p2p

    if not p2p.enabled():
        return None
    g = p2p.default_group()
    # a wait of an earlier collective that ran out (a late or dead peer) surfaces here, before more work is
    # queued on NaN-filled results: P2PTimeout
    g.poll()
    return g if x.numel() * 4 <= g.max_bytes else None


def check_p2p() -> None:
    """Synchronise and raise ``P2PTimeout`` if a peer-to-peer collective of this process has timed out (workflows
    call it where they synchronise anyway: monitors' host reads, checkpoints)."""
    from ..parallel import p2p

    if p2p.enabled() and p2p._default is not None:
        p2p._default.check()


def _via_host(x: torch.Tensor) -> bool:
    # gloo collectives are given CPU tensors only; device data is staged through host memory
    return x.is_cuda and _dist().get_backend() == "gloo"


def all_gather(x: torch.Tensor, axis_name: Optional[str] = POP_AXIS_NAME, axis: int = 0, tiled: bool = True,
               group=None) -> torch.Tensor:
    """Concatenate (``tiled``) or stack every rank's ``x`` along ``axis``.

    Identity when ``axis_name`` is None or the job has one process.
    """
    if axis_name is None or not is_dist_initialized() or get_world_size() == 1:
        return x if tiled else x.unsqueeze(axis)
    d = _dist()
    w = d.get_world_size(group)
    src = x.contiguous()
    if group is None and axis == 0 and tiled and src.dim() >= 1:
        p2p_group = _p2p_group(src)
        if p2p_group is not None:
            return p2p_group.all_gather(src)
    stage = _via_host(src)
    if stage:
        src = src.cpu()
    if axis == 0:
        out = src.new_empty((w * src.shape[0],) + tuple(src.shape[1:])) if src.dim() else src.new_empty((w,))
        d.all_gather_into_tensor(out, src.reshape(-1) if src.dim() == 0 else src, group=group)
        parts = out if tiled and src.dim() else out.reshape((w,) + tuple(src.shape))
        res = parts
    else:
        bufs = [torch.empty_like(src) for _ in range(w)]
        d.all_gather(bufs, src, group=group)
        res = torch.cat(bufs, dim=axis) if tiled else torch.stack(bufs, dim=axis)
    return res.to(x.device) if stage else res


def all_reduce(x: torch.Tensor, op: str = "sum", group=None) -> torch.Tensor:
    """Out-of-place all-reduce (sum / max / min / mean); identity in a single-process job."""
    if not is_dist_initialized() or get_world_size() == 1:
        return x
    d = _dist()
    ops = {"sum": d.ReduceOp.SUM, "max": d.ReduceOp.MAX, "min": d.ReduceOp.MIN, "mean": d.ReduceOp.SUM}
    if group is None and op in ("sum", "mean") and x.dtype == torch.float32:
        p2p_group = _p2p_group(x)
        if p2p_group is not None:
            y = p2p_group.all_reduce(x.detach())
            return y / d.get_world_size() if op == "mean" else y
    y = x.detach().clone().contiguous()
    stage = _via_host(y)
    if stage:
        y = y.cpu()
    d.all_reduce(y, op=ops[op], group=group)
    if op == "mean":
        y = y / d.get_world_size(group)
    return y.to(x.device) if stage else y


def all_reduce_(x: torch.Tensor, group=None) -> torch.Tensor:
    """Sum all-reduce that may overwrite ``x`` (a message buffer the caller owns) and returns the
    reduced tensor: no defensive copy on the RCCL path."""
    if not is_dist_initialized() or get_world_size() == 1:
        return x
    p2p_group = _p2p_group(x) if (group is None and x.dtype == torch.float32) else None
    if p2p_group is not None:
        return p2p_group.all_reduce(x)
    if _via_host(x):
        y = x.cpu()
        _dist().all_reduce(y, group=group)
        return y.to(x.device)
    x = x if x.is_contiguous() else x.contiguous()
    _dist().all_reduce(x, group=group)
    return x


class _Reduced:
    """Handle of :func:`all_reduce_begin`: ``wait()`` returns the reduced tensor (on the RCCL path it makes the current
    stream wait for the collective; the host does not block)."""

    def __init__(self, result, work=None):
        self._result, self._work = result, work

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._result


def all_reduce_begin(x: torch.Tensor, group=None) -> _Reduced:
    """Start the sum all-reduce of the message buffer ``x`` (overwritten, like :func:`all_reduce_`) and return a handle.
    On the RCCL path the collective runs on the communicator's stream: kernels issued on the current stream before
    ``wait()`` — work that does not depend on the result — overlap with it.  The other paths (single process, host
    staging for gloo, peer-to-peer kernels) complete here and hand back a finished handle."""
    if not is_dist_initialized() or get_world_size() == 1:
        return _Reduced(x)
    p2p_group = _p2p_group(x) if (group is None and x.dtype == torch.float32) else None
    if p2p_group is not None or _via_host(x) or not x.is_cuda:
        return _Reduced(all_reduce_(x, group))
    x = x if x.is_contiguous() else x.contiguous()
    return _Reduced(x, _dist().all_reduce(x, group=group, async_op=True))


def broadcast(x: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if not is_dist_initialized() or get_world_size() == 1:
        return x
    y = x.detach().clone().contiguous()
    stage = _via_host(y)
    if stage:
        y = y.cpu()
    _dist().broadcast(y, src=src, group=group)
    return y.to(x.device) if stage else y


def barrier(group=None) -> None:
    if is_dist_initialized() and get_world_size() > 1:
        _dist().barrier(group=group)


def tree_all_gather(tree, axis_name: Optional[str] = POP_AXIS_NAME, **kwargs):
    from .tree import tree_map

    return tree_map(lambda t: all_gather(t, axis_name, **kwargs), tree)


def unpmap(x, axis_name: Optional[str] = None):
    """Replica 0 of a value that carries a leading replica axis (no-op otherwise).

    Kept for API parity: in this framework every process holds its own un-batched replica, so
    states never grow a device axis.
    """
    return x


def tree_unpmap(tree, axis_name: Optional[str] = None):
    return tree
