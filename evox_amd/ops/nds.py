"""Pareto ranking and crowding distance.

Host path (also the oracle of the kernels): dominance matrix + front peeling in PyTorch.
GPU path: ``csrc/nds.hip`` — matrix-free tiled dominance counting and per-front peeling
(compacted index lists, counts in registers/LDS), segmented sorts for crowding distance.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native
from ._util import pick, check, hip, on_gpu, stream_of


def _has(symbol: str) -> bool:
    try:
        getattr(hip(), symbol)
        return True
    except AttributeError:
        return False


def non_dominated_rank_reference(x: torch.Tensor, until: Optional[int] = None) -> torch.Tensor:
    n = x.shape[0]
    xf = x.float()
    a, b = xf[:, None, :], xf[None, :, :]
    dom = (a <= b).all(-1) & (a < b).any(-1)                      # dom[i, j]: i dominates j
    count = dom.sum(0).to(torch.int64)
    rank = torch.zeros(n, dtype=torch.int64, device=x.device)
    assigned = torch.zeros(n, dtype=torch.bool, device=x.device)
    cur = 0
    front = count == 0
    domf = dom.to(torch.float32)
    while bool(front.any()):
        rank = torch.where(front, torch.full_like(rank, cur), rank)
        assigned |= front
        count = count - (front.to(torch.float32) @ domf).to(torch.int64) - front.to(torch.int64)
        cur += 1
        if until is not None and int(assigned.sum()) >= until:
            rank = torch.where(assigned, rank, torch.full_like(rank, cur))
            break
        front = count == 0
    else:
        # rows that can never reach count 0 (NaN objectives): ranked last
        rank = torch.where(assigned, rank, torch.full_like(rank, cur))
    return rank


def non_dominated_rank(x: torch.Tensor, until: Optional[int] = None) -> torch.Tensor:
    if on_gpu(x):
        from . import _nds_hip

        if _nds_hip.supported(x):
            return _nds_hip.rank(x, until)
    n = x.shape[0]
    if n > 8192:
        return _blocked_reference(x, until)
    return non_dominated_rank_reference(x, until)


def _blocked_reference(x: torch.Tensor, until: Optional[int]) -> torch.Tensor:
    """Same result as the reference without materialising n×n at once (row blocks)."""
    n = x.shape[0]
    xf = x.float()
    blk = 2048
    count = torch.zeros(n, dtype=torch.int64, device=x.device)
    for s in range(0, n, blk):
        a = xf[s:s + blk, None, :]
        count += ((a <= xf[None]).all(-1) & (a < xf[None]).any(-1)).sum(0)
    rank = torch.zeros(n, dtype=torch.int64, device=x.device)
    assigned = torch.zeros(n, dtype=torch.bool, device=x.device)
    cur = 0
    front = count == 0
    while bool(front.any()):
        rank[front] = cur
        assigned |= front
        f = xf[front]
        rem = ~assigned
        if bool(rem.any()):
            r = xf[rem]
            dec = torch.zeros(r.shape[0], dtype=torch.int64, device=x.device)
            for s in range(0, f.shape[0], blk):
                a = f[s:s + blk, None, :]
                dec += ((a <= r[None]).all(-1) & (a < r[None]).any(-1)).sum(0)
            count[rem] -= dec
        count[front] = -1
        cur += 1
        if until is not None and int(assigned.sum()) >= until:
            break
        front = (count == 0) & ~assigned
    rank[~assigned] = cur
    return rank


def crowding_distance_reference(costs: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, m = costs.shape
    dev = costs.device
    if mask is None:
        mask = torch.ones(n, dtype=torch.bool, device=dev)
    nvalid = mask.sum()
    c = costs.float()
    # masked-out rows sort last in every objective
    keyed = torch.where(mask[:, None], c, torch.full_like(c, float("inf")))
    order = torch.argsort(keyed, dim=0, stable=True)                       # (n, m)
    sorted_c = torch.gather(keyed, 0, order)
    last = (nvalid - 1).clamp_min(0)
    span = pick(sorted_c, last) - sorted_c[0]                              # (m,)
    gaps = torch.zeros_like(sorted_c)
    gaps[1:-1] = (sorted_c[2:] - sorted_c[:-2]) / span
    pos = torch.arange(n, device=dev)[:, None]
    gaps = torch.where((pos == 0) | (pos == last), torch.full_like(gaps, float("inf")), gaps)
    dist_sorted = gaps
    dist = torch.zeros_like(c)
    dist.scatter_(0, order, dist_sorted)
    total = dist.sum(1)
    return torch.where(mask, total, torch.full_like(total, float("-inf")))


def crowding_distance(costs: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    return crowding_distance_reference(costs, mask)


def segmented_crowding_distance(costs: torch.Tensor, group: torch.Tensor, mask: torch.Tensor, n_groups: int) -> torch.Tensor:
    """``crowding_distance`` of several disjoint sets at once: among the rows selected by ``mask``, those with
    the same ``group`` (int64 in ``[0, n_groups)``) form one set with its own extremes and spans.  Two stable
    sorts per objective for all sets together; rows outside ``mask`` get ``-inf``."""
    n, m = costs.shape
    dev = costs.device
    c = costs.float()
    g = torch.where(mask, group, torch.full_like(group, n_groups))
    total = torch.zeros(n, device=dev)
    spare = torch.full_like(g, n_groups + 1)
    span_all = None
    if n < 8192:
        # few rows: one pair of atomic reductions for all objectives costs less than two scatters per objective
        where = g[:, None].expand(n, m)
        inf = torch.full((n_groups + 1, m), float("inf"), device=dev)
        span_all = (-inf).scatter_reduce(0, where, c, reduce="amax") - inf.scatter_reduce(0, where, c, reduce="amin")
    for k in range(m):
        o = torch.argsort(c[:, k], stable=True)
        o = o[torch.argsort(g[o], stable=True)]                          # grouped, ascending inside a group
        gs, v = g[o], c[o, k]
        edge = torch.ones(n, dtype=torch.bool, device=dev)
        edge[1:] = gs[1:] != gs[:-1]
        tail = torch.ones(n, dtype=torch.bool, device=dev)
        tail[:-1] = edge[1:]
        # a set's extremes are the first and the last value of its segment: two plain scatters with one writer per set
        # (everything else goes to a spare row) instead of atomic min / max reductions of all rows onto a few addresses
        if span_all is None:
            lo = torch.zeros(n_groups + 2, device=dev).scatter_(0, torch.where(edge, gs, spare), v)
            hi = torch.zeros(n_groups + 2, device=dev).scatter_(0, torch.where(tail, gs, spare), v)
            span = hi - lo
        else:
            span = span_all[:, k]
        gap = (torch.roll(v, -1) - torch.roll(v, 1)) / span[gs]
        gap = torch.where(edge | tail, torch.full_like(gap, float("inf")), gap)
        total[o] += gap
    return torch.where(mask, total, torch.full_like(total, float("-inf")))


_CROWD_KERNEL_MAX = 1 << 22      # front sizes up to which the neighbour-reading crowding kernel is used


def rank_crowding_select(fit: torch.Tensor, k: int) -> torch.Tensor:
    """Indices of the ``k`` survivors of NSGA-II truncation: whole fronts in order of their index
    (population order inside a front), then the members of the front that straddles the cut by
    decreasing crowding distance (ties: population order).

    On the GPU the front sizes come back from the peeling loop, so only the straddling front is
    touched afterwards: its members are gathered, their crowding distances come from a sort-free
    kernel and one small counting sort picks the most isolated ones — two sorts of the merged
    population in total instead of five."""
    n = fit.shape[0]
    if on_gpu(fit):
        from . import _nds_hip
        from . import sort as _sort

        if _nds_hip.supported(fit) and k < n and torch.cuda.is_current_stream_capturing():
            # recording into a HIP graph: nothing may come back to the host.  Populations that the one-launch sort takes
            # (merged size <= 1024) are ranked without a read-back and truncated by tensor ops; larger ones need the
            # peeling loop's progress on the host and cannot be recorded
            if n > _nds_hip._SMALL_MAX_N:
                raise RuntimeError(f"a non-dominated sort of {n} > {_nds_hip._SMALL_MAX_N} points reads its progress "
                                   "back to the host and cannot be recorded into a graph")
            return _truncate_from_rank(fit, _nds_hip.rank(fit, until=k), k)
        if _nds_hip.supported(fit) and k < n:
            rank, (cut, n_before, cut_size) = _nds_hip.rank(fit, until=k, with_info=True)
            need = k - n_before
            if 0 < need <= cut_size <= _CROWD_KERNEL_MAX:
                order = _sort.argsort_small_ints(rank, cut + 1)         # front indices: one radix pass per byte
                head = order[:n_before]
                members = order[n_before:n_before + cut_size]          # the cut front, in population order
                dist = _nds_hip.crowding(fit[members])
                chosen = members[_sort.argsort(-dist)[:need]]
                return torch.cat([head, chosen])
            return _truncate_from_rank(fit, rank, k)
    rank = non_dominated_rank(fit, until=k)
    return _truncate_from_rank(fit, rank, k)


def _truncate_from_rank(fit, rank, k):
    order = torch.argsort(rank, stable=True)
    worst = pick(rank, order[k - 1])
    mask = rank == worst
    dist = crowding_distance(fit, mask)
    key2 = torch.where(mask, -dist, torch.zeros_like(dist))
    idx = torch.argsort(key2, stable=True)
    return idx[torch.argsort(rank[idx], stable=True)][:k]
