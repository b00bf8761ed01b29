"""Top-k, roulette-wheel, uniform-random and random-p-best selection (reference:
``src/evox/operators/selection/{topk_fit,roulette_wheel,uniform_random,find_pbest}.py``)."""
from __future__ import annotations

import torch

from ...core import random as rnd
from .non_dominate import _dedup_first


def topk_fit(population, fitness, topk, deduplicate=False):
    key = fitness
    if deduplicate:
        key = torch.where(_dedup_first(population), fitness, torch.full_like(fitness, float("inf")))
    index = torch.argsort(key, stable=True)[:topk]
    return population[index], fitness[index]


class TopkFit:
    def __init__(self, topk, deduplicate=False):
        self.topk = topk
        self.deduplicate = deduplicate

    def __call__(self, population, fitness):
        return topk_fit(population, fitness, self.topk, self.deduplicate)


class RouletteWheelSelection:
    """Probability ∝ 1 / (shifted fitness): smaller fitness, higher chance."""

    def __init__(self, n):
        self.n = n

    def __call__(self, key, x, fitness):
        fit = fitness - torch.clamp(fitness.min(), max=0.0) + 1e-6
        cdf = torch.cumsum(1.0 / fit, 0)
        cdf = cdf / cdf[-1]
        u = rnd.uniform(key, (self.n,))
        idx = torch.searchsorted(cdf, u).clamp_(max=x.shape[0] - 1)
        return x[idx], idx


def uniform_rand(key, pop, *others, prob):
    num = int(pop.shape[0] * prob)
    chosen = rnd.randint(key, (num,), 0, pop.shape[0])
    if not others:
        return pop[chosen]
    return (pop[chosen], *[o[chosen] for o in others])


class UniformRand:
    def __init__(self, prob):
        self.prob = prob

    def __call__(self, key, pop, *others):
        return uniform_rand(key, pop, *others, prob=self.prob)


def select_rand_pbest(key, percent, population, fitness, num=None):
    """A random member of the best ``percent`` of the population.  ``num=None`` returns one
    vector (reference signature); ``num=k`` returns ``k`` independent picks (k, dim) with the
    top-p set computed once instead of once per individual."""
    if not 0 < percent <= 1.0:
        raise ValueError("percent must be in (0, 1]")
    n = population.shape[0]
    top = max(int(n * percent), 1)
    from ...ops.sort import select_indices

    best = select_indices(fitness, top)                   # the top set, in index order
    pick = rnd.randint(key, (1 if num is None else num,), 0, top)
    rows = population[best[pick]]
    return rows[0] if num is None else rows
