"""Sorting helpers (argsort / top-k / rank / companion-array sort).

On the GPU, fp32 key vectors up to ``_RANK_SORT_MAX`` long are sorted by the one-launch
counting-rank kernel (``csrc/sort.hip``); longer or non-fp32 inputs go to the tensor library."""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({"evox_hip_rank_sort": (I, [P, I, P, P, P]), "evox_hip_select": (I, [P, I, I, I, P, P])})

_RANK_SORT_MAX = 16384


def _rank_sortable(keys: torch.Tensor) -> bool:
    return keys.is_cuda and keys.dtype == torch.float32 and keys.dim() == 1 and 0 < keys.shape[0] <= _RANK_SORT_MAX


def _rank_sort(keys: torch.Tensor, want_rank: bool):
    n = keys.shape[0]
    keys = keys.contiguous()
    order = torch.empty(n, dtype=torch.int64, device=keys.device)
    rank = torch.empty(n, dtype=torch.int64, device=keys.device) if want_rank else None
    check(hip().evox_hip_rank_sort(keys.data_ptr(), n, order.data_ptr(), rank.data_ptr() if want_rank else None,
                                   stream_of(keys)), "rank_sort")
    return order, rank


_SELECT_MAX_EAGER = 1 << 16   # measured (µs, kernel vs library top-k): 10 000 keys 24 vs 37, 50 000 keys 78 vs 49
_SELECT_MAX_GRAPH = 1 << 14


def select_indices(keys: torch.Tensor, k: int, largest: bool = False) -> torch.Tensor:
    """Indices (int64) of the ``k`` smallest — or largest — entries of the 1-D ``keys``: a SET, for callers that
    draw from it or scatter into it.  NaN entries are taken last, ties at the threshold go to the lower index; the
    indices come in ascending order except on the library route described below, which leaves them unordered.

    On the GPU: a radix select in ONE launch by one persistent workgroup (``csrc/sort.hip::select_kernel``).  Its
    device time grows with 1.3 ns per key (the instruction issue of a single compute unit), so beyond 16 384 keys
    the library's multi-kernel top-k is faster on the device — but it costs eight launches, and an eagerly issued
    generation of a DE variant is bound by launches.  Hence the kernel is used up to 65 536 keys when work is
    issued eagerly and up to 16 384 while a HIP graph is being recorded (launches are free at replay, and sorting
    the library's result by index would cost more than it saves); the two routes return the same set unless
    several keys tie exactly at the threshold."""
    n = keys.shape[0]
    k = int(k)
    if not 1 <= k <= n:
        raise ValueError("k must be in [1, n]")
    if keys.is_cuda and keys.dtype == torch.float32 and keys.dim() == 1:
        limit = _SELECT_MAX_GRAPH if torch.cuda.is_current_stream_capturing() else _SELECT_MAX_EAGER
        if n <= limit:
            keys = keys.contiguous()
            out = torch.empty(k, dtype=torch.int64, device=keys.device)
            check(hip().evox_hip_select(keys.data_ptr(), n, k, int(bool(largest)), out.data_ptr(), stream_of(keys)), "select")
            return out
        clean = torch.where(torch.isnan(keys), torch.full_like(keys, float("-inf") if largest else float("inf")), keys)
        return torch.topk(clean, k, largest=largest, sorted=False).indices
    clean = torch.where(torch.isnan(keys), torch.full_like(keys, float("-inf") if largest else float("inf")), keys)
    # stable order: ties at the threshold go to the lower index, as in the kernel
    order = torch.argsort(-clean if largest else clean, stable=True)[:k]
    return torch.sort(order).values


def sort_by_key(keys: torch.Tensor, *vals):
    """Sort ``keys`` ascending and permute every companion array the same way
    (reference: ``src/evox/algorithms/so/es_variants/sort_utils.py:5-12``)."""
    order = argsort(keys)
    return (keys[order], *[v[order] for v in vals])


def argsort(keys: torch.Tensor) -> torch.Tensor:
    """Stable ascending argsort (NaN last)."""
    if _rank_sortable(keys):
        return _rank_sort(keys, False)[0]
    return torch.argsort(keys, stable=True)


def argsort_with_rank(keys: torch.Tensor):
    """``(order, rank)`` with ``rank[order[r]] = r`` (the position of every key in the sorted order)."""
    if _rank_sortable(keys):
        return _rank_sort(keys, True)
    order = torch.argsort(keys, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.shape[0], device=order.device)
    return order, rank
