"""Quality indicators for multi-objective optimisation: GD, GD+, IGD, IGD+ and Monte-Carlo
hypervolume (reference: ``src/evox/metrics/{gd,igd,hypervolume}.py``)."""
from __future__ import annotations

import torch

from ..core import random as rnd


def _pairwise(a, b):
    from ..ops.distance import cdist

    return cdist(a.float(), b.float())


def _pairwise_plus(a, b):
    """Modified distance of GD+/IGD+: only the components where ``a`` is worse than ``b``."""
    d = (a[:, None, :] - b[None, :, :]).clamp_min(0)
    return torch.sqrt((d * d).sum(-1))


def gd(objs, pf, p=1):
    """Generational distance: mean distance from each obtained point to the true front."""
    d = _pairwise(objs, pf).amin(1)
    return torch.linalg.vector_norm(d, ord=p) / d.shape[0]


def gd_plus(objs, pf, p=1):
    d = _pairwise_plus(objs, pf).amin(1)
    return torch.linalg.vector_norm(d, ord=p) / d.shape[0]


def igd(objs, pf, p=1):
    """Inverted generational distance: mean distance from each front point to the obtained set."""
    d = _pairwise(pf, objs).amin(1)
    return torch.linalg.vector_norm(d, ord=p) / d.shape[0]


def igd_plus(objs, pf, p=1):
    d = _pairwise_plus(objs, pf).amin(0)
    return torch.linalg.vector_norm(d, ord=p) / d.shape[0]


class _FrontMetric:
    fn = None

    def __init__(self, pf, p=1):
        self.pf, self.p = pf, p

    def __call__(self, objs):
        return type(self).fn(objs, self.pf.to(objs.device), self.p)


class GD(_FrontMetric):
    fn = staticmethod(gd)


class GDPlus(_FrontMetric):
    fn = staticmethod(gd_plus)


class IGD(_FrontMetric):
    fn = staticmethod(igd)


class IGDPlus(_FrontMetric):
    fn = staticmethod(igd_plus)


class HV:
    """Monte-Carlo hypervolume with respect to reference point ``ref`` (minimisation).

    ``sample_method="bounding_cube"`` samples the box between the ideal point of the set and
    ``ref`` and counts dominated samples; ``"each_cube"`` samples inside each point's own box and
    weights every sample by the inverse of the number of boxes containing it."""

    def __init__(self, ref, num_sample=100_000, sample_method="bounding_cube"):
        if sample_method not in ("bounding_cube", "each_cube"):
            raise ValueError("sample_method must be 'bounding_cube' or 'each_cube'")
        self.ref = torch.as_tensor(ref, dtype=torch.float32)
        self.num_sample = num_sample
        self.sample_method = sample_method

    def __call__(self, key, objs):
        from ..ops import hv as _hv

        ref = self.ref.to(objs.device)
        if self.sample_method == "bounding_cube":
            return _hv.hypervolume_bounding_cube(key, objs, ref, self.num_sample)
        return _hv.hypervolume_each_cube(key, objs, ref, self.num_sample)
