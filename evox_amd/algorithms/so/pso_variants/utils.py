"""Small helpers of the PSO family (reference: ``pso_variants/utils.py:8-60``)."""
from __future__ import annotations

import torch

from ....core import random as rnd
from ....utils.common import min_by  # noqa: F401  (re-exported)


def get_distance_matrix(location: torch.Tensor) -> torch.Tensor:
    """Euclidean distance between every pair of rows."""
    from ....ops.distance import cdist

    return cdist(location, location)


def row_argsort(x: torch.Tensor) -> torch.Tensor:
    return torch.argsort(x, dim=-1, stable=True)


def select_from_mask(key, mask: torch.Tensor, s: int) -> torch.Tensor:
    """0/1 vector with ``s`` ones placed uniformly at random among the ones of ``mask``."""
    n = mask.shape[0]
    noise = rnd.uniform(key, (n,))
    order = torch.argsort(mask.to(noise.dtype) + noise, descending=True)
    out = torch.zeros_like(mask)
    out[order[:s]] = 1
    return out


def init_swarm(key_pop, key_vel, lb, ub, pop_size, mean=None, stdev=None):
    """Uniform (or Gaussian, when ``mean``/``stdev`` are given) swarm and velocities."""
    shape = (pop_size, lb.shape[0])
    if mean is not None and stdev is not None:
        population = torch.clamp(stdev * rnd.normal(key_pop, shape) + mean, lb, ub)
        velocity = stdev * rnd.normal(key_vel, shape)
    else:
        length = ub - lb
        population = rnd.uniform(key_pop, shape) * length + lb
        velocity = (rnd.uniform(key_vel, shape) * 2 - 1) * length
    return population, velocity


def update_bests(population, fitness, pbest, pbest_fit, gbest, gbest_fit):
    """Personal and global best bookkeeping shared by the family."""
    better = fitness < pbest_fit
    pbest = torch.where(better[:, None], population, pbest)
    pbest_fit = torch.minimum(pbest_fit, fitness)
    best_fit, i = torch.min(fitness, dim=0)                   # 0-d tensors; no index is read back to the host
    take = best_fit < gbest_fit.reshape(())
    gbest = torch.where(take, population.index_select(0, i.reshape(1))[0], gbest)
    gbest_fit = torch.where(take, best_fit, gbest_fit.reshape(())).reshape(gbest_fit.shape)
    return pbest, pbest_fit, gbest, gbest_fit
