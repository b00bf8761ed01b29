"""Sampling of weight / design points: Das–Dennis simplex lattice (with Deb–Jain's two-layer
variant), Latin hypercube and grid (reference: ``src/evox/operators/sampling/{uniform,
latin_hypercude,grid}.py``).  The lattice is small host-side combinatorics; the result is a
tensor on the requested device."""
from __future__ import annotations

from itertools import combinations
from math import comb

import torch

from ...core import random as rnd


def _simplex_lattice(h: int, m: int) -> torch.Tensor:
    """All points of the (m-1)-simplex with coordinates that are multiples of 1/h."""
    c = torch.tensor(list(combinations(range(1, h + m), m - 1)), dtype=torch.float64)
    c = c - torch.arange(m - 1, dtype=torch.float64) - 1
    upper = torch.cat([c, torch.full((c.shape[0], 1), float(h), dtype=torch.float64)], dim=1)
    lower = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.float64), c], dim=1)
    return (upper - lower) / h


class UniformSampling:
    """``UniformSampling(n, m)() -> (w, n_actual)``: at most ``n`` points on the unit simplex in
    ``m`` dimensions.  ``n_actual`` generally differs from ``n`` (largest lattice that fits)."""

    def __init__(self, n=None, m=None, device=None):
        self.n, self.m, self.device = n, m, device

    def __call__(self, key=None):
        n, m = self.n, self.m
        h1 = 1
        while comb(h1 + m, m - 1) <= n:
            h1 += 1
        w = _simplex_lattice(h1, m)
        if h1 < m:
            h2 = 0
            while comb(h1 + m - 1, m - 1) + comb(h2 + m, m - 1) <= n:
                h2 += 1
            if h2 > 0:
                w2 = _simplex_lattice(h2, m)
                w = torch.cat([w, w2 / 2.0 + 1.0 / (2.0 * m)], dim=0)
        w = w.clamp_min(1e-6).to(torch.float32)
        if self.device is not None:
            w = w.to(self.device)
        return w, w.shape[0]


class LatinHypercubeSampling:
    """``n`` points in the unit cube of dimension ``d``; each axis is stratified into n cells."""

    def __init__(self, n=None, d=None, smooth=True):
        self.n, self.d, self.smooth = n, d, smooth

    def __call__(self, key):
        k_perm, k_jit = rnd.split(key)
        perm = rnd.batched_permutation(k_perm, self.d, self.n).T.to(torch.float32)     # (n, d)
        if self.smooth:
            return (perm + rnd.uniform(k_jit, (self.n, self.d))) / self.n
        return (perm + 0.5) / self.n


class GridSampling:
    """Regular grid with ``ceil(n ** (1/m))`` points per axis: returns ``(w, n_actual)``."""

    def __init__(self, n=None, m=None):
        self.n, self.m = n, m

    def __call__(self, key=None):
        gap = torch.linspace(0, 1, int(round(self.n ** (1 / self.m) + 0.4999999)))
        grids = torch.meshgrid(*([gap] * self.m), indexing="ij")
        w = torch.stack([g.reshape(-1) for g in grids], dim=1)
        return w, w.shape[0]
