"""Simulated binary crossover in the PlatEMO convention: parents are the first and second
halves of the population (reference: ``src/evox/operators/crossover/sbx.py:7-61``)."""
from __future__ import annotations

import torch

from ...core import random as rnd


def simulated_binary(key, x, pro_c=1.0, dis_c=20.0, type=1):
    n, d = x.shape
    h = n // 2
    p1, p2 = x[:h], x[h:2 * h]
    k_mu, k_sign, k_gene, k_pair = rnd.split(key, 4)
    mu = rnd.uniform(k_mu, (h, d))
    beta = torch.where(mu <= 0.5, (2 * mu) ** (1 / (dis_c + 1)), (2 - 2 * mu) ** (-1 / (dis_c + 1)))
    beta = beta * (1 - 2 * rnd.randint(k_sign, (h, d), 0, 2).to(beta.dtype))
    beta = torch.where(rnd.uniform(k_gene, (h, d)) < 0.5, torch.ones_like(beta), beta)
    beta = torch.where(rnd.uniform(k_pair, (h, 1)) > pro_c, torch.ones_like(beta), beta)
    mid, half = (p1 + p2) / 2, (p1 - p2) / 2
    if type == 1:
        off = torch.cat([mid + beta * half, mid - beta * half], dim=0)
        if n % 2:
            off = torch.cat([off, x[-1:]], dim=0)
        return off
    if type == 2:
        return mid + beta * half
    raise ValueError("type must be 1 or 2")


class SimulatedBinary:
    """``pro_c`` crossover probability per pair, ``dis_c`` distribution index; ``type=1``
    produces as many offspring as parents, ``type=2`` half as many."""

    def __init__(self, pro_c=1, dis_c=20, type=1):
        self.pro_c, self.dis_c, self.type = pro_c, dis_c, type

    def __call__(self, key, x):
        return simulated_binary(key, x, self.pro_c, self.dis_c, self.type)
