"""Swarm communication topologies as adjacency matrices / padded neighbour lists
(reference: ``pso_variants/topology_utils.py:14-251``).  Topologies are built once per run (or
rarely), so this is plain tensor code; the per-generation consumers work on the padded
``(N, K)`` neighbour lists, K = maximum degree, instead of N×N structures."""
from __future__ import annotations

import math

import torch

from ....core import random as rnd
from .utils import get_distance_matrix, row_argsort, select_from_mask


def build_adjacancy_matrix_by_K_nearest_neighbour(distance_ranking: torch.Tensor, K: int) -> torch.Tensor:
    """Symmetric 0/1 matrix linking every node with its K nearest nodes (and itself)."""
    n = distance_ranking.shape[0]
    rows = torch.arange(n, device=distance_ranking.device)[:, None].expand(n, K + 1).reshape(-1)
    cols = distance_ranking[:, : K + 1].reshape(-1)
    a = torch.zeros((n, n), dtype=torch.int64, device=distance_ranking.device)
    a[rows, cols] = 1
    a[cols, rows] = 1
    return a


def mutate_shortcut(key, adjacancy_matrix: torch.Tensor, num_shortcut: int) -> torch.Tensor:
    """Rewire ``num_shortcut`` undirected edges: one endpoint of each chosen edge moves to a
    random node (small-world shortcuts)."""
    if num_shortcut <= 0:
        return adjacancy_matrix
    n = adjacancy_matrix.shape[0]
    dev = adjacancy_matrix.device
    k1, k2, k3 = rnd.split(key, 3)
    eye = torch.eye(n, dtype=adjacancy_matrix.dtype, device=dev)
    flat = torch.triu(adjacancy_matrix - eye).reshape(-1)
    rows = torch.arange(n, device=dev).repeat_interleave(n)
    cols = torch.arange(n, device=dev).repeat(n)
    chosen = select_from_mask(k1, flat, num_shortcut).bool()
    move_row = rnd.bernoulli(k2, 0.5, (n * n,))
    node = rnd.randint(k3, (n * n,), 0, n)
    rows = torch.where(chosen & move_row, node, rows)
    cols = torch.where(chosen & ~move_row, node, cols)
    out = torch.zeros_like(adjacancy_matrix)
    out.index_put_((rows, cols), flat, accumulate=True)
    return torch.clamp(out + out.T + eye, 0, 1)


def get_circles_neighbour(random_key, population: torch.Tensor, K: int = 2, shortcut: int = 0) -> torch.Tensor:
    """K-nearest-neighbour graph in decision space with ``shortcut`` rewired edges."""
    ranking = row_argsort(get_distance_matrix(population))
    a = build_adjacancy_matrix_by_K_nearest_neighbour(ranking, K)
    return mutate_shortcut(random_key, a, shortcut)


def get_full_neighbour(population: torch.Tensor) -> torch.Tensor:
    n = population.shape[0]
    return torch.ones((n, n), dtype=torch.int64, device=population.device)


def get_ring_neighbour(population: torch.Tensor, K: int = 1) -> torch.Tensor:
    """Node i is linked to (i ± j) mod n for j = 0..K."""
    n = population.shape[0]
    i = torch.arange(n, device=population.device)
    d = (i[:, None] - i[None, :]).abs()
    d = torch.minimum(d, n - d)
    return (d <= K).to(torch.int64)


def get_square_neighbour(population: torch.Tensor) -> torch.Tensor:
    """Von-Neumann neighbourhood on a torus lattice whose sides are as equal as N allows."""
    n = population.shape[0]
    col = math.floor(math.sqrt(n))
    while col > 1 and n % col != 0:
        col -= 1
    row = n // col
    idx = torch.arange(n).reshape(col, row)
    a = torch.zeros((n, n), dtype=torch.int64)
    for di, dj in ((0, 1), (1, 0), (0, -1), (-1, 0)):
        nb = torch.roll(idx, shifts=(-di, -dj), dims=(0, 1))
        a[idx.reshape(-1), nb.reshape(-1)] = 1
    return a.to(population.device)


def build_adjacancy_list_from_matrix(adjacancy_matrix: torch.Tensor, keep_self_loop: bool = True, width=None):
    """Padded neighbour lists: ``(list (N, K), mask (N, K))`` with K the maximum degree (or
    ``width``); padding entries point at the node itself and have mask 0."""
    a = adjacancy_matrix != 0
    n = a.shape[0]
    if not keep_self_loop:
        a = a & ~torch.eye(n, dtype=torch.bool, device=a.device)
    k = int(a.sum(1).max()) if width is None else width
    k = max(k, 1)
    # stable sort puts the neighbours (True) first, in index order
    order = torch.argsort((~a).to(torch.int8), dim=1, stable=True)[:, :k]
    mask = torch.gather(a, 1, order)
    me = torch.arange(n, device=a.device)[:, None].expand(n, k)
    return torch.where(mask, order, me), mask.to(adjacancy_matrix.dtype)


def get_neighbour_best_fitness(fitness: torch.Tensor, adjacancy_list: torch.Tensor):
    """Best fitness among each node's listed neighbours and whose it is."""
    f = fitness[adjacancy_list]
    j = torch.argmin(f, dim=1)
    best_idx = torch.gather(adjacancy_list, 1, j[:, None])[:, 0]
    return fitness[best_idx], best_idx
