"""Mutation operators (reference: ``src/evox/operators/mutation/{pm_mutation,bitflip,
gaussian}.py``)."""
from __future__ import annotations

import torch

from ...core import random as rnd


def polynomial(key, x, boundary, pro_m=1, dis_m=20):
    """PlatEMO polynomial mutation: each gene mutates with probability ``pro_m / d``; the step
    follows the polynomial distribution of index ``dis_m`` and respects the bounds.  For an odd
    number of rows (>1) the last row is passed through unchanged."""
    k_site, k_mu = rnd.split(key)
    odd = x.shape[0] > 1 and x.shape[0] % 2 == 1
    pop = x if x.shape[0] == 1 else x[: (x.shape[0] // 2) * 2]
    n, d = pop.shape
    lower, upper = boundary[0], boundary[1]
    site = rnd.uniform(k_site, (n, d)) < pro_m / d
    mu = rnd.uniform(k_mu, (n, d))
    pop = torch.maximum(torch.minimum(pop, upper), lower)
    span = upper - lower
    lo_branch = site & (mu <= 0.5)
    hi_branch = site & (mu > 0.5)
    norm_lo = (pop - lower) / span
    norm_hi = (upper - pop) / span
    e = dis_m + 1.0
    d_lo = (2.0 * mu + (1.0 - 2.0 * mu) * (1.0 - norm_lo).clamp_min(0) ** e).clamp_min(0) ** (1.0 / e) - 1.0
    d_hi = 1.0 - (2.0 * (1.0 - mu) + 2.0 * (mu - 0.5) * (1.0 - norm_hi).clamp_min(0) ** e).clamp_min(0) ** (1.0 / e)
    out = torch.where(lo_branch, pop + span * d_lo, pop)
    out = torch.where(hi_branch, pop + span * d_hi, out)
    if odd:
        out = torch.cat([out, x[-1:]], dim=0)
    return out


class Polynomial:
    def __init__(self, boundary, pro_m=1, dis_m=20):
        self.boundary = boundary
        self.pro_m, self.dis_m = pro_m, dis_m

    def __call__(self, key, x):
        return polynomial(key, x, self.boundary, self.pro_m, self.dis_m)


def _packbits(b: torch.Tensor) -> torch.Tensor:
    n = b.shape[-1]
    pad = (-n) % 8
    if pad:
        b = torch.cat([b, torch.zeros(b.shape[:-1] + (pad,), dtype=b.dtype, device=b.device)], dim=-1)
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device=b.device)
    return (b.reshape(b.shape[:-1] + (-1, 8)).to(torch.uint8) * w).sum(-1).to(torch.uint8)


def bitflip(key, x, prob):
    """Flip each bit with probability ``prob``: bool input is XOR-ed with a random mask; uint8
    input is treated as packed bits (8 per byte)."""
    if x.dtype == torch.uint8:
        mask = rnd.uniform(key, x.shape + (8,)) < prob
        return x ^ _packbits(mask).reshape(x.shape)
    mask = rnd.uniform(key, x.shape) < prob
    return x ^ mask if x.dtype == torch.bool else torch.logical_xor(x.bool(), mask).to(x.dtype)


class Bitflip:
    def __init__(self, prob):
        self.prob = prob

    def __call__(self, key, x):
        return bitflip(key, x, self.prob)


def gaussian(key, x, stdvar):
    return x + stdvar * rnd.normal(key, x.shape)


class Gaussian:
    def __init__(self, stdvar=1.0):
        self.stdvar = stdvar

    def __call__(self, key, x):
        return gaussian(key, x, self.stdvar)
