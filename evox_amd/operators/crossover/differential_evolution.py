"""Differential-evolution variation operators, batched over the individuals being varied.

The reference defines these per individual and maps them over the population
(``src/evox/operators/crossover/differential_evolution.py:10-123``); here every function takes
the whole batch so that one call is a handful of kernels.
"""
from __future__ import annotations

import torch

from ...core import random as rnd


def differential_evolve(key, x1, x2, x3, F, CR):
    """DE/rand/1 with uniform crossover against ``x1``."""
    mutant = x1 + F * (x2 - x3)
    mask = rnd.uniform(key, x1.shape) < CR
    return torch.where(mask, mutant, x1)


class DifferentialEvolve:
    def __init__(self, F=0.5, CR=1):
        self.F, self.CR = F, CR

    def __call__(self, key, p1, p2, p3):
        return differential_evolve(key, p1, p2, p3, self.F, self.CR)


def de_diff_sum(key, diff_padding_num, num_diff_vects, index, population, replace=False):
    """Signed sum of ``num_diff_vects`` difference vectors of random individuals.

    ``index`` is the vector (B,) of individuals being mutated; ``num_diff_vects`` an int or a
    (B,) tensor (individuals may use different counts, padded to ``diff_padding_num`` draws).
    Returns ``(difference_sum (B, dim), rand_vect_idx (B,))`` where ``rand_vect_idx`` is one
    more random individual usable as the base vector.  Individuals are drawn without
    replacement per row (``replace=False``) and never equal ``index``."""
    n, dim = population.shape
    index = torch.as_tensor(index, device=population.device).reshape(-1)
    b = index.shape[0]
    if replace:
        choice = rnd.randint(key, (b, diff_padding_num), 0, n)
    else:
        # per-row sampling without replacement: the diff_padding_num smallest random words
        w = rnd.bits(key, (b, n))
        choice = torch.topk(w, diff_padding_num, dim=1, largest=False).indices
    choice = torch.where(choice == index[:, None], torch.full_like(choice, n - 1), choice)
    nd = torch.as_tensor(num_diff_vects, device=population.device).reshape(-1, 1)
    select_len = nd * 2 + 1
    pos = torch.arange(diff_padding_num, device=population.device)[None, :]
    live = (pos < select_len) & (pos >= 1)
    sign = torch.where(pos % 2 == 1, 1.0, -1.0) * live                      # +x1 -x2 +x3 -x4 ...
    diff = torch.einsum("bk,bkd->bd", sign.to(population.dtype), population[choice])
    return diff, choice[:, 0]


def de_bin_cross(key, mutation_vector, current_vect, CR):
    """Binomial crossover with one forced dimension per row; ``CR`` scalar or (B,)."""
    k_r, k_m = rnd.split(key)
    b, dim = mutation_vector.shape
    jrand = rnd.randint(k_r, (b, 1), 0, dim)
    cr = CR.to(mutation_vector.dtype).reshape(-1, 1) if isinstance(CR, torch.Tensor) else float(CR)
    mask = rnd.uniform(k_m, (b, dim)) < cr
    mask = mask | (torch.arange(dim, device=mask.device)[None, :] == jrand)
    return torch.where(mask, mutation_vector, current_vect)


def de_exp_cross(key, mutation_vector, current_vect, CR):
    """Exponential crossover: a run of geometric length starting at a random dimension."""
    k_n, k_l = rnd.split(key)
    b, dim = mutation_vector.shape
    start = rnd.randint(k_n, (b, 1), 0, dim)
    if isinstance(CR, torch.Tensor):
        cr = CR.to(torch.float32).reshape(-1).expand(b)
    else:
        cr = torch.full((b,), float(CR), dtype=torch.float32, device=mutation_vector.device)
    length = torch.clamp(rnd.geometric(k_l, cr), max=dim) - 1
    pos = (torch.arange(dim, device=start.device)[None, :] - start) % dim
    mask = pos < length[:, None]
    return torch.where(mask, mutation_vector, current_vect)


def de_arith_recom(mutation_vector, current_vect, K):
    k = K
    if isinstance(K, torch.Tensor):
        k = K.to(mutation_vector.dtype)
        if k.dim() == 1:
            k = k[:, None]
    return current_vect + k * (mutation_vector - current_vect)
