"""One-shot collectives over peer-mapped device memory (``csrc/p2p.hip``).

``P2PGroup`` gives the ranks of one node an ``all_gather`` and a sum ``all_reduce`` in which every
rank writes its contribution directly into every peer's receive buffer (all xGMI links at once) and
then waits — for a bounded time — for the peers' flags.  It needs a ``torch.distributed`` process
group only to exchange the IPC handles at construction.

Sums of at least ``two_shot_min_bytes`` go in two shots (reduce-scatter, then all-gather, both direct), so each
link carries 2/W of the message instead of all of it.

A wait that runs out of polls fills its output with NaNs and sets an error word that the host can read without
synchronising (a pinned copy): every later call of the group — and ``check()`` — raises :class:`P2PTimeout`.
After a time-out the group is out of step with its peers and must not be used again.

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
    "evox_hip_p2p_put": (I, [P, I64, P, I, I, I64, I64, P, P, I64, P]),
    "evox_hip_p2p_collect": (I, [P, I, I64, I64, I64, P, P, P, I, ctypes.c_longlong, P, P, P]),
    "evox_hip_p2p_two_shot_finish": (I, [P, P, I, I, I64, I64, I64, I64, I64, I64, P, P, P, ctypes.c_longlong,
                                         P, P, P, P]),
    "evox_hip_host_device_ptr": (I, [P, P]),
})


class P2PTimeout(RuntimeError):
    """A peer's contribution did not arrive within the polling budget."""


class P2PGroup:
    def __init__(self, max_bytes: int = 8 << 20, group=None, device: Optional[torch.device] = None,
                 timeout_s: float = 20.0, two_shot_min_bytes: Optional[int] = None):
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
        # reduced chunks of the two-shot sum: one chunk per rank and slot
        self.chunk_stride = ((self.max_bytes + self.world - 1) // self.world + 63) // 64 * 64
        self._red_offset = self._header + 2 * self.slot_stride
        total = self._red_offset + 2 * self.world * self.chunk_stride
        self.two_shot_min_bytes = int(os.environ.get("EVOX_AMD_P2P_TWO_SHOT_MIN", str(256 << 10))) \
            if two_shot_min_bytes is None else int(two_shot_min_bytes)
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
        self._done2 = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the same word in pinned host memory, written by the wait kernels: readable without a synchronisation
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        dev_ptr = ctypes.c_void_p()
        check(lib.evox_hip_host_device_ptr(self._err_host.data_ptr(), ctypes.byref(dev_ptr)), "hipHostGetDevicePointer")
        self._err_host_dev = dev_ptr.value
        self._broken = False
        # sequence number of the collectives, kept on the device and advanced by the last kernel of each one: the
        # launches of a collective take no argument that changes from call to call (they can be recorded into a graph)
        self._seq_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._done_c = torch.zeros(1, dtype=torch.int32, device=self.device)
        # ≈ 1 µs per poll (a sleep of 32·64 cycles plus the load): the budget bounds the wait kernel
        self._max_polls = max(1000, int(timeout_s * 1e6))
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
        self._raise_if_timed_out()
        out = torch.empty(n if reduce else n * self.world, dtype=src.dtype, device=self.device)
        lib, s = hip(), stream_of(src)
        two_shot = reduce and self.world > 1 and n * 4 >= self.two_shot_min_bytes
        # words of a chunk: a multiple of 4 (16-byte stores) that covers the message with W chunks
        chunk = ((n + self.world - 1) // self.world + 3) // 4 * 4 if two_shot else 0
        seq, done_c = self._seq_dev.data_ptr(), self._done_c.data_ptr()
        check(lib.evox_hip_p2p_put(src.data_ptr(), n, self._peer_table.data_ptr(), self.world, self.rank,
                                   self.slot_stride, self.max_bytes, seq, self._done.data_ptr(), chunk, s), "p2p_put")
        if two_shot:
            check(lib.evox_hip_p2p_two_shot_finish(self._own, self._peer_table.data_ptr(), self.world, self.rank,
                                                   self.slot_stride, self.max_bytes, self._red_offset, self.chunk_stride,
                                                   chunk, n, seq, done_c, out.data_ptr(), self._max_polls,
                                                   self._done2.data_ptr(), self._err.data_ptr(), self._err_host_dev, s),
                  "p2p_two_shot_finish")
        else:
            check(lib.evox_hip_p2p_collect(self._own, self.world, self.slot_stride, self.max_bytes, n, seq, done_c,
                                           out.data_ptr(), int(reduce), self._max_polls, self._err.data_ptr(),
                                           self._err_host_dev, s), "p2p_collect")
        return out

    def _raise_if_timed_out(self) -> None:
        if self._broken or int(self._err_host[0]):
            self._broken = True
            raise P2PTimeout("a peer's message did not arrive in time; the outputs of that collective are NaN-filled "
                             "and this group is out of step with its peers")

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """Concatenation of every rank's ``x`` along dim 0 (all ranks pass the same shape)."""
        out = self._run(x, False)
        return out.reshape((self.world * x.shape[0],) + tuple(x.shape[1:])) if x.dim() else out

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """Element-wise sum over ranks (fp32), out of place."""
        return self._run(x, True).reshape(x.shape)

    def check(self) -> None:
        """Synchronise and raise ``P2PTimeout`` if any wait of this group ran out of polls."""
        if int(self._err.item()):
            self._broken = True
        self._raise_if_timed_out()

    def poll(self) -> None:
        """The same without synchronising: sees the time-outs of the collectives that have finished by now."""
        self._raise_if_timed_out()

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
