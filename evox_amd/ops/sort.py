"""Sorting helpers (argsort / top-k / rank / companion-array sort).

On the GPU, fp32 key vectors up to ``_RANK_SORT_MAX`` long are sorted by the one-launch
counting-rank kernel (``csrc/sort.hip``); longer or non-fp32 inputs go to the tensor library."""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({"evox_hip_rank_sort": (I, [P, I, P, P, P]),
                          "evox_hip_rank_sort_scalar": (I, [P, I, P, P, P]),
                          "evox_hip_rank_sort_bucket": (I, [P, I, P, P, P]),
                          "evox_hip_rank_sort_scratch": (_native.I64, [I]),
                          "evox_hip_rank_sort_sliced": (I, [P, I, P, P, P, P]),
                          "evox_hip_rank_sort_values": (I, [P, I, P, P, P, P, P, P]), "evox_hip_select": (I, [P, I, I, I, P, P]),
                          "evox_hip_select_reg": (I, [P, I, I, I, P, P]),
                          "evox_hip_select_grid_ws": (_native.I64, []), "evox_hip_select_grid_max": (I, []),
                          "evox_hip_select_grid": (I, [P, I, I, I, P, P, I, P]),
                          "evox_hip_select_wide_ws": (_native.I64, [I]),
                          "evox_hip_select_wide": (I, [P, I, I, I, P, P, P])})

_native.register_hip_api({"evox_hip_radix_ws_words": (_native.I64, [_native.I64]),
                          "evox_hip_radix_argsort": (I, [P, I, _native.I64, P, I, I, I, P, P, P, P])})

_RANK_SORT_MAX = 16384
_RADIX_MAX = 1 << 21          # the scatter pass of the radix sort sums a 256 x tiles table per workgroup: tiles <= 512


def radix_sortable(keys: torch.Tensor) -> bool:
    return (keys.is_cuda and keys.dim() == 1 and keys.dtype in (torch.float32, torch.int32)
            and 0 < keys.shape[0] <= _RADIX_MAX)


def radix_argsort(keys: torch.Tensor, perm: torch.Tensor = None, bits=(0, 32), want_rank: bool = False):
    """Stable ascending argsort of a long 1-D key vector by the hand-written radix sort (``csrc/radix.hip``): fp32 keys
    (NaN last, -0 = +0; any element stride, e.g. a column of a matrix) or int32 keys.  ``perm`` (int64): the sort
    permutes this index sequence instead of ``0..n-1`` and gathers the keys through it — the stable sort by a more
    significant column after a sort by a less significant one (lexicographic and segmented sorts).  ``bits``: the key
    bits that take part (multiples of 8; for small non-negative integers the low bytes suffice).  Returns ``order``
    (int64), or ``(order, rank)`` with ``rank[order[r]] = r``."""
    n = keys.shape[0]
    dev = keys.device
    mode = 1 if keys.dtype == torch.float32 else 2
    if perm is not None:
        perm = perm.contiguous()
        if perm.dtype != torch.int64 or perm.shape[0] != n:
            raise ValueError("perm must be int64 of the keys' length")
    order = torch.empty(n, dtype=torch.int64, device=dev)
    rank = torch.empty(n, dtype=torch.int64, device=dev) if want_rank else None
    ws = torch.empty(int(hip().evox_hip_radix_ws_words(n)), dtype=torch.int32, device=dev)
    check(hip().evox_hip_radix_argsort(keys.data_ptr(), mode, keys.stride(0), perm.data_ptr() if perm is not None else None,
                                       n, int(bits[0]), int(bits[1]), order.data_ptr(),
                                       rank.data_ptr() if want_rank else None, ws.data_ptr(), stream_of(keys)),
          "radix_argsort")
    return (order, rank) if want_rank else order


def argsort_small_ints(v: torch.Tensor, max_value: int) -> torch.Tensor:
    """Stable ascending argsort of non-negative integers bounded by ``max_value`` (front indices, group numbers): on
    the GPU one radix pass per byte of the bound."""
    n = v.shape[0]
    if v.is_cuda and v.dim() == 1 and 16 < n <= _RADIX_MAX and 0 <= max_value < (1 << 31):
        bits = 8 if max_value < (1 << 8) else (16 if max_value < (1 << 16) else (24 if max_value < (1 << 24) else 32))
        return radix_argsort(v.to(torch.int32), bits=(0, bits))
    return torch.argsort(v, stable=True)


def lexsort_rows(x: torch.Tensor) -> torch.Tensor:
    """Stable argsort of the rows of ``x`` (n, m) in lexicographic order, first column most significant: a chain of
    single-column radix sorts from the last column on (the tensor library's three argsorts and gathers otherwise)."""
    n, m = x.shape
    if x.is_cuda and x.dtype == torch.float32 and 16 < n <= _RADIX_MAX:
        order = None
        for k in range(m - 1, -1, -1):
            order = radix_argsort(x[:, k], order)
        return order
    order = torch.argsort(x[:, m - 1], stable=True)
    for k in range(m - 2, -1, -1):
        order = order[torch.argsort(x[order, k], stable=True)]
    return order

_SLICED_MIN = 2048          # from here on the two-launch sliced count beats the one-launch form


def _rank_sortable(keys: torch.Tensor) -> bool:
    return keys.is_cuda and keys.dtype == torch.float32 and keys.dim() == 1 and 0 < keys.shape[0] <= _RANK_SORT_MAX


def _rank_sort(keys: torch.Tensor, want_rank: bool, form: str = None):
    import os

    n = keys.shape[0]
    keys = keys.contiguous()
    order = torch.empty(n, dtype=torch.int64, device=keys.device)
    rank = torch.empty(n, dtype=torch.int64, device=keys.device) if want_rank else None
    form = form or os.environ.get("EVOX_AMD_RANK_SORT", "scalar")
    if form in ("bucket", "scalar"):
        # one launch: the plain count with the keys in scalar registers (17 µs at 10⁴ keys), or the rank by splitters
        # (n log n compares, but its redundant per-workgroup partition costs more than it saves up to 16 384 keys: 21 µs)
        fn = hip().evox_hip_rank_sort_bucket if form == "bucket" else hip().evox_hip_rank_sort_scalar
        check(fn(keys.data_ptr(), n, order.data_ptr(), rank.data_ptr() if want_rank else None, stream_of(keys)),
              "rank_sort_" + form)
        return order, rank
    if n >= _SLICED_MIN:
        # too few 64-row blocks to fill the chip: the count is spread over four times as many workgroups
        scratch = torch.empty(int(hip().evox_hip_rank_sort_scratch(n)), dtype=torch.int32, device=keys.device)
        check(hip().evox_hip_rank_sort_sliced(keys.data_ptr(), n, order.data_ptr(), rank.data_ptr() if want_rank else None,
                                              scratch.data_ptr(), stream_of(keys)), "rank_sort_sliced")
        return order, rank
    check(hip().evox_hip_rank_sort(keys.data_ptr(), n, order.data_ptr(), rank.data_ptr() if want_rank else None,
                                   stream_of(keys)), "rank_sort")
    return order, rank


# Time per call (µs, back-to-back on one stream) at 10 000 / 50 000 / 262 144 keys: one workgroup with the keys in
# registers 15 / 53 / -, one launch of co-operating workgroups (with the fill of its workspace) 22 / 24 / 29, one
# workgroup re-reading the keys 22 / 76 / 327, the five-launch form 42 / 44 / 52, the library's top-k 37 / 50 / 122
# (tools/select_timing.py).  SHADE at 50 000 individuals under graph replay (island workflow): 0.163 -> 0.144 ms per generation.
_SELECT_REG_MAX = 1 << 14
_SELECT_ROUTE = None                   # tests: force "registers" / "grid" / "one_workgroup" / "wide"


def _select_route(n: int) -> str:
    if _SELECT_ROUTE is not None:
        return _SELECT_ROUTE
    if n <= _SELECT_REG_MAX:
        return "registers"
    if n <= int(hip().evox_hip_select_grid_max()):
        return "grid"
    return "wide"


def select_grid(keys: torch.Tensor, k: int, largest: bool = False, spin_limit: int = 0):
    """The one-launch chip-wide selection and its workspace (int32; word 1 counts workgroups whose bounded wait at a
    grid-wide barrier ran out — the output is then zero-filled instead of a selection)."""
    n = keys.shape[0]
    out = torch.empty(k, dtype=torch.int64, device=keys.device)
    ws = torch.zeros((int(hip().evox_hip_select_grid_ws()) + 3) // 4, dtype=torch.int32, device=keys.device)
    check(hip().evox_hip_select_grid(keys.data_ptr(), n, k, int(bool(largest)), out.data_ptr(), ws.data_ptr(),
                                     int(spin_limit), stream_of(keys)), "select_grid")
    return out, ws


def select_indices(keys: torch.Tensor, k: int, largest: bool = False) -> torch.Tensor:
    """Indices (int64, ascending) of the ``k`` smallest — or largest — entries of the 1-D ``keys``; NaN entries are
    taken last, ties at the threshold go to the lower index.

    On the GPU a radix select (``csrc/sort.hip``) instead of a sort, always ONE launch up to 262 144 keys: three
    histogram passes over 11 + 11 + 10 bits of the keys' integer image, then an ordered write-out, with the keys held
    in registers throughout — by one workgroup up to 16 384 keys, above that by up to 64 workgroups that meet at
    grid-wide barriers.  Longer vectors: the same passes as five launches whose last-finishing workgroup scans the
    global histogram / the per-block counts."""
    n = keys.shape[0]
    k = int(k)
    if not 1 <= k <= n:
        raise ValueError("k must be in [1, n]")
    if keys.is_cuda and keys.dtype == torch.float32 and keys.dim() == 1:
        keys = keys.contiguous()
        route = _select_route(n)
        if route == "grid":
            return select_grid(keys, k, largest)[0]
        out = torch.empty(k, dtype=torch.int64, device=keys.device)
        args = (keys.data_ptr(), n, k, int(bool(largest)), out.data_ptr())
        if route == "registers":
            check(hip().evox_hip_select_reg(*args, stream_of(keys)), "select_reg")
        elif route == "one_workgroup":
            check(hip().evox_hip_select(*args, stream_of(keys)), "select")
        else:
            ws = torch.zeros((int(hip().evox_hip_select_wide_ws(n)) + 3) // 4, dtype=torch.int32, device=keys.device)
            check(hip().evox_hip_select_wide(*args, ws.data_ptr(), stream_of(keys)), "select_wide")
        return out
    clean = torch.where(torch.isnan(keys), torch.full_like(keys, float("-inf") if largest else float("inf")), keys)
    # stable order: ties at the threshold go to the lower index, as in the kernels
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
    if radix_sortable(keys):
        return radix_argsort(keys)
    return torch.argsort(keys, stable=True)


def sort_with_rank(keys: torch.Tensor, out: torch.Tensor = None, copy_out: torch.Tensor = None):
    """``(sorted_keys, rank)``: the keys in ascending order (NaN last) and the stable rank of every key.  ``out``
    (contiguous, same shape and dtype) receives the sorted keys, ``copy_out`` (likewise) an unsorted copy of the keys."""
    def fits(t):
        return t is None or (t.is_contiguous() and t.dtype == keys.dtype and t.shape == keys.shape)

    if _rank_sortable(keys) and fits(out) and fits(copy_out):
        n = keys.shape[0]
        keys = keys.contiguous()
        rank = torch.empty(n, dtype=torch.int64, device=keys.device)
        out = torch.empty_like(keys) if out is None else out
        scratch = torch.empty(int(hip().evox_hip_rank_sort_scratch(n)), dtype=torch.int32, device=keys.device)
        check(hip().evox_hip_rank_sort_values(keys.data_ptr(), n, None, rank.data_ptr(), out.data_ptr(),
                                              copy_out.data_ptr() if copy_out is not None else None, scratch.data_ptr(),
                                              stream_of(keys)), "rank_sort_values")
        return out, rank
    if radix_sortable(keys) and keys.dtype == torch.float32 and fits(out) and fits(copy_out):
        order, rank = radix_argsort(keys, want_rank=True)
        values = keys[order]
        if out is not None:
            out.copy_(values)
            values = out
        if copy_out is not None:
            copy_out.copy_(keys)
        return values, rank
    values, order = torch.sort(keys, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(keys.shape[0], device=keys.device)
    if out is not None:
        out.copy_(values)
        values = out
    if copy_out is not None:
        copy_out.copy_(keys)
    return values, rank


def argsort_with_rank(keys: torch.Tensor):
    """``(order, rank)`` with ``rank[order[r]] = r`` (the position of every key in the sorted order)."""
    if _rank_sortable(keys):
        return _rank_sort(keys, True)
    if radix_sortable(keys):
        return radix_argsort(keys, want_rank=True)
    order = torch.argsort(keys, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.shape[0], device=order.device)
    return order, rank
