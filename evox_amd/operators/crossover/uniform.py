"""Uniform and one-point crossover on randomly paired parents (reference:
``crossover/uniform.py:6-37``, ``crossover/one_point.py:6-38``)."""
from __future__ import annotations

import torch

from ...core import random as rnd
from .paired_sbx import _random_pairing, _unpair


def uniform_rand(key, parents):
    k_pair, k_mask = rnd.split(key)
    p1, p2, perm = _random_pairing(k_pair, parents)
    mask = rnd.uniform(k_mask, p1.shape) < 0.5
    c1 = torch.where(mask, p1, p2)
    c2 = torch.where(mask, p2, p1)
    return _unpair(c1, c2, parents, perm)


class UniformRand:
    """Each gene of a pair is swapped with probability one half."""

    def __call__(self, key, parents):
        return uniform_rand(key, parents)


def one_point(key, parents):
    k_pair, k_cut = rnd.split(key)
    p1, p2, perm = _random_pairing(k_pair, parents)
    d = parents.shape[1]
    cut = rnd.randint(k_cut, (p1.shape[0], 1), 0, d + 1)
    mask = torch.arange(d, device=parents.device)[None, :] < cut
    c1 = torch.where(mask, p1, p2)
    c2 = torch.where(mask, p2, p1)
    return _unpair(c1, c2, parents, perm)


class OnePoint:
    def __call__(self, key, parents):
        return one_point(key, parents)
