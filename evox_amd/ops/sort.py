"""Sorting helpers (argsort / top-k / rank / companion-array sort).

On the GPU, fp32 key vectors up to ``_RANK_SORT_MAX`` long are sorted by the one-launch
counting-rank kernel (``csrc/sort.hip``); longer or non-fp32 inputs go to the tensor library."""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({"evox_hip_rank_sort": (I, [P, I, P, P, P])})

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
