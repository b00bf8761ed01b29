"""Driver of the matrix-free non-dominated sort kernels (``csrc/nds.hip``)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({
    "evox_hip_nds_count": (I, [P, _native.I64, I, I, P, I, P]),
    "evox_hip_nds_round": (I, [P, _native.I64, I, I, P, P, P, P, P, P, I, P]),
})

MAX_OBJECTIVES = 8
_ROUNDS_PER_CHECK = 4


def supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dim() == 2 and 2 <= x.shape[1] <= MAX_OBJECTIVES and x.shape[0] > 0


def rank(x: torch.Tensor, until: Optional[int] = None) -> torch.Tensor:
    """Front index of every row (int64).  With ``until`` peeling stops after the front that brings
    the number of ranked points to at least ``until``; all later points share the next index."""
    n, m = x.shape
    dev = x.device
    # ascending first objective: a dominator never sits behind the point it dominates
    order = torch.argsort(x[:, 0].float(), stable=True)
    ld = (n + 15) // 16 * 16
    obj = torch.full((m, ld), float("inf"), dtype=torch.float32, device=dev)          # objective-major, padded columns
    obj[:, :n] = x.float()[order].T
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    blocks = (n + 255) // 256
    ysplit = max(1, min(32, 1024 // blocks))
    s = stream_of(obj)
    lib = hip()
    check(lib.evox_hip_nds_count(obj.data_ptr(), ld, n, m, count.data_ptr(), ysplit, s), "nds_count")

    rk = torch.full((n,), -1, dtype=torch.int32, device=dev)
    front = torch.empty((m, ld), dtype=torch.float32, device=dev)                  # objectives of the current front
    sizes = torch.zeros(n + _ROUNDS_PER_CHECK, dtype=torch.int32, device=dev)     # size of front k
    blockcnt = torch.empty(2 * blocks, dtype=torch.int32, device=dev)             # per-block counts | running totals
    target = n if until is None else min(int(until), n)
    cur = 0
    while True:
        for _ in range(_ROUNDS_PER_CHECK):
            check(lib.evox_hip_nds_round(obj.data_ptr(), ld, n, m, count.data_ptr(), rk.data_ptr(), front.data_ptr(),
                                         sizes[cur:].data_ptr(), blockcnt.data_ptr(), blockcnt[blocks:].data_ptr(),
                                         cur, s), "nds_round")
            cur += 1
        done = sizes[:cur].cumsum(0)
        host = done.cpu()                                   # one small read-back per batch of rounds
        if int(host[-1]) >= target or cur >= n or int(host[-1]) == int(host[-_ROUNDS_PER_CHECK - 1] if cur > _ROUNDS_PER_CHECK else 0):
            break
    host_sizes = host.tolist()
    # the front at which the target was reached; everything later collapses into one index
    cut = next((k for k, c in enumerate(host_sizes) if c >= target), cur - 1)
    while cut > 0 and host_sizes[cut] == host_sizes[cut - 1]:
        cut -= 1                                            # trailing empty rounds
    rk64 = rk.to(torch.int64)
    last = cut + 1
    rk64 = torch.where((rk64 < 0) | (rk64 > cut), torch.full_like(rk64, last), rk64)
    if until is None and int(host_sizes[-1]) == n:
        pass
    out = torch.empty_like(rk64)
    out[order] = rk64
    return out
