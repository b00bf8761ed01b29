"""Non-dominated sorting, crowding distance and NSGA-II survivor selection.

``non_dominated_sort`` assigns each point the index of its Pareto front (0 = non-dominated).
On a GPU it runs the matrix-free tiled kernels of ``evox_amd.ops.nds`` (no N×N matrix is ever
stored); on the host it peels fronts from the dominance matrix.  Reference behaviour:
``src/evox/operators/selection/non_dominate.py:32-232``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ...ops import nds as _nds


def non_dominated_sort(x: torch.Tensor, method: str = "auto", until: Optional[int] = None) -> torch.Tensor:
    """Front index of every row of ``x`` (n, m); rows containing NaN are never ranked ahead of
    anything.  ``until``: stop once at least that many points are ranked — the remaining points
    all get the next front index (enough for truncation selection, and much cheaper).
    ``method`` is accepted for API compatibility ("auto", "scan", "full map-reduce", "host")."""
    if method not in ("auto", "scan", "full map-reduce", "host"):
        raise ValueError("method must be 'auto', 'scan', 'full map-reduce' or 'host'")
    return _nds.non_dominated_rank(x, until=until)


def crowding_distance(costs: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NSGA-II crowding distance of the rows selected by ``mask`` (others get ``-inf``):
    per objective, neighbours' gap normalised by the objective's range, ±inf at the extremes,
    summed over objectives."""
    return _nds.crowding_distance(costs, mask)


def crowding_distance_sort(x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Indices that sort ``x`` by ascending crowding distance."""
    return torch.argsort(crowding_distance(x, mask), stable=True)


def _dedup_first(population: torch.Tensor):
    """Indices of the first occurrence of every distinct row, then the duplicates."""
    _, inverse = torch.unique(population, dim=0, return_inverse=True)
    n = population.shape[0]
    ar = torch.arange(n, device=population.device)
    first = torch.full((int(inverse.max()) + 1,), n, device=population.device, dtype=torch.long)
    first = first.scatter_reduce(0, inverse, ar, reduce="amin")
    is_first = first[inverse] == ar
    return is_first


def non_dominate(population, fitness, topk: int, deduplicate: bool = False):
    """Keep the ``topk`` best rows by (front index, then larger crowding distance within the
    last admitted front).  With ``deduplicate`` repeated decision vectors compete with infinite
    fitness (their true fitness is returned if they are selected anyway)."""
    fit_for_rank = fitness
    if deduplicate:
        keep = _dedup_first(population)
        fit_for_rank = torch.where(keep[:, None], fitness, torch.full_like(fitness, float("inf")))
    # composite order: front index first, larger crowding distance first inside the cut front
    idx = _nds.rank_crowding_select(fit_for_rank, topk)
    return population[idx], fitness[idx]


class NonDominate:
    def __init__(self, topk, deduplicate=False):
        self.topk = topk
        self.deduplicate = deduplicate

    def __call__(self, population, fitness):
        return non_dominate(population, fitness, self.topk, self.deduplicate)
