"""SBX on randomly paired parents (reference: ``crossover/simulated_binary.py:6-61``)."""
from __future__ import annotations

import torch

from ...core import random as rnd


def _random_pairing(key, x):
    n = x.shape[0]
    perm = rnd.permutation(key, n)
    half = n // 2
    return x[perm[:half]], x[perm[half:2 * half]], perm


def _unpair(c1, c2, x, perm):
    half = c1.shape[0]
    out = x.clone()
    out[perm[:half]] = c1
    out[perm[half:2 * half]] = c2
    return out


def sbx(key, x, distribution_factor=1, stdvar=1.0):
    """Pairs are drawn by a random permutation; each pair is blended gene by gene with the SBX
    spread factor β (distribution index ``distribution_factor``) scaled by ``stdvar``."""
    k_pair, k_u = rnd.split(key)
    p1, p2, perm = _random_pairing(k_pair, x)
    u = rnd.uniform(k_u, p1.shape)
    eta = distribution_factor
    beta = torch.where(u <= 0.5, (2 * u) ** (1 / (eta + 1)), (1 / (2 - 2 * u).clamp_min(1e-12)) ** (1 / (eta + 1)))
    beta = beta * stdvar
    c1 = 0.5 * ((1 + beta) * p1 + (1 - beta) * p2)
    c2 = 0.5 * ((1 - beta) * p1 + (1 + beta) * p2)
    return _unpair(c1, c2, x, perm)


class SBXCrossover:
    def __init__(self, stdvar=1.0, distribution_factor=1):
        self.stdvar = stdvar
        self.distribution_factor = distribution_factor

    def __call__(self, key, x):
        return sbx(key, x, self.distribution_factor, self.stdvar)
