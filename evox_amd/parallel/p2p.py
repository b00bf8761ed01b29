"""One-shot collectives over peer-mapped device memory (``csrc/p2p.hip``).

``P2PGroup`` gives the ranks of one node an ``all_gather`` and a sum ``all_reduce`` in which every
rank writes its contribution directly into every peer's receive buffer (all xGMI links at once) and
then waits — for a bounded time — for the peers' flags.  It needs a ``torch.distributed`` process
group only to exchange the IPC handles at construction.

Opt-in: ``core.distributed`` routes device tensors of at most ``max_bytes`` through a group when
``EVOX_AMD_P2P=1``; RCCL stays the default.  The path has been exercised with several processes
sharing one GPU (the only multi-process GPU configuration available to its tests), not yet across
xGMI links.  The reference's single collective is ``all_gather`` of the fitness slices
(``src/evox/workflows/std_workflow.py:160``).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from .. import _native
from ..ops._util import check, hip, stream_of

P, I, I64 = _native.P, _native.I, _native.I64
_native.register_hip_api({
    "evox_hip_p2p_header_bytes": (I, []),
    "evox_hip_p2p_max_world": (I, []),
    "evox_hip_p2p_handle_bytes": (I, []),
    "evox_hip_p2p_alloc": (I, [P, I64]),
    "evox_hip_p2p_free": (I, [P]),
    "evox_hip_p2p_export": (I, [P, P, I]),
    "evox_hip_p2p_import": (I, [P, P]),
    "evox_hip_p2p_close": (I, [P]),
    "evox_hip_p2p_put": (I, [P, I64, P, I, I, I, I64, I64, ctypes.c_uint32, P, P]),
    "evox_hip_p2p_collect": (I, [P, I, I, I64, I64, I64, ctypes.c_uint32, P, I, ctypes.c_longlong, P, P]),
})


class P2PTimeout(RuntimeError):
    """A peer's contribution did not arrive within the polling budget."""


class P2PGroup:
    def __init__(self, max_bytes: int = 8 << 20, group=None, device: Optional[torch.device] = None,
                 timeout_s: float = 20.0):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("P2PGroup needs an initialised torch.distributed process group to exchange handles")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        lib = hip()
        if self.world > lib.evox_hip_p2p_max_world():
            raise ValueError(f"at most {lib.evox_hip_p2p_max_world()} ranks")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_bytes = (int(max_bytes) + 63) // 64 * 64
        self.slot_stride = self.world * self.max_bytes
        self._header = lib.evox_hip_p2p_header_bytes()
        total = self._header + 2 * self.slot_stride
        with torch.cuda.device(self.device):
            own = ctypes.c_void_p()
            check(lib.evox_hip_p2p_alloc(ctypes.byref(own), total), "p2p_alloc")
            self._own = own.value
            hb = lib.evox_hip_p2p_handle_bytes()
            handle = (ctypes.c_char * hb)()
            check(lib.evox_hip_p2p_export(self._own, handle, hb), "p2p_export (hipIpcGetMemHandle)")
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle), group=group)
            self._peers = []
            self._opened = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    self._peers.append(self._own)
                    continue
                ptr = ctypes.c_void_p()
                buf = (ctypes.c_char * hb).from_buffer_copy(h)
                check(lib.evox_hip_p2p_import(buf, ctypes.byref(ptr)), "p2p_import (hipIpcOpenMemHandle)")
                self._peers.append(ptr.value)
                self._opened.append(ptr.value)
        self._peer_table = torch.tensor(self._peers, dtype=torch.int64, device=self.device)
        self._done = torch.zeros(self.world, dtype=torch.int32, device=self.device)
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._seq = 0
        # ≈ 1 µs per poll (a sleep of 32·64 cycles plus the load): the budget bounds the wait kernel
        self._max_polls = max(1000, int(timeout_s * 1e6))
        self._checked = 0
        dist.barrier(group=group)                     # every buffer is mapped before anyone writes

    # ------------------------------------------------------------------------------------
    def _run(self, x: torch.Tensor, reduce: bool) -> torch.Tensor:
        if not x.is_cuda or x.device != self.device:
            raise ValueError("P2PGroup works on tensors of its own device")
        if reduce and x.dtype != torch.float32:
            raise TypeError("all_reduce sums fp32 tensors")
        if x.element_size() != 4:
            raise TypeError("4-byte element types only (fp32 / int32)")
        src = x.contiguous()
        n = src.numel()
        if n == 0:
            return src.clone() if reduce else src.new_empty((0,))
        if n * 4 > self.max_bytes:
            raise ValueError(f"message of {n * 4} bytes exceeds the group's max_bytes={self.max_bytes}")
        self._seq = (self._seq + 1) & 0xFFFFFFFF or 1
        slot = self._seq & 1
        out = torch.empty(n if reduce else n * self.world, dtype=src.dtype, device=self.device)
        lib, s = hip(), stream_of(src)
        check(lib.evox_hip_p2p_put(src.data_ptr(), n, self._peer_table.data_ptr(), self.world, self.rank, slot,
                                   self.slot_stride, self.max_bytes, self._seq, self._done.data_ptr(), s), "p2p_put")
        check(lib.evox_hip_p2p_collect(self._own, self.world, slot, self.slot_stride, self.max_bytes, n, self._seq,
                                       out.data_ptr(), int(reduce), self._max_polls, self._err.data_ptr(), s),
              "p2p_collect")
        return out

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """Concatenation of every rank's ``x`` along dim 0 (all ranks pass the same shape)."""
        out = self._run(x, False)
        return out.reshape((self.world * x.shape[0],) + tuple(x.shape[1:])) if x.dim() else out

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """Element-wise sum over ranks (fp32), out of place."""
        return self._run(x, True).reshape(x.shape)

    def check(self) -> None:
        """Synchronise and raise ``P2PTimeout`` if any wait since the last check ran out of polls."""
        if int(self._err.item()):
            self._err.zero_()
            raise P2PTimeout("a peer's message did not arrive in time")

    def close(self) -> None:
        lib = hip()
        torch.cuda.synchronize(self.device)
        for ptr in self._opened:
            lib.evox_hip_p2p_close(ptr)
        self._opened = []
        if self._own:
            lib.evox_hip_p2p_free(self._own)
            self._own = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default: Optional[P2PGroup] = None


def enabled() -> bool:
    return os.environ.get("EVOX_AMD_P2P", "0") == "1"


def default_group() -> P2PGroup:
    global _default
    if _default is None:
        _default = P2PGroup(max_bytes=int(os.environ.get("EVOX_AMD_P2P_MAX_BYTES", str(8 << 20))))
    return _default
