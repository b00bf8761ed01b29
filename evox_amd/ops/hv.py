"""Hypervolume: Monte-Carlo estimators (S samples × N points × m dominance tests, evaluated in
sample chunks so that the comparison table never exceeds a few tens of MB) and an exact
slicing algorithm for small fronts.

All routines work on the boxes spanned by each point and the reference point,
``|objs - ref|``, so they serve minimisation (reference above the front) and maximisation
(reference below it) alike, as the reference's ``metrics/hypervolume.py:9-12`` does."""
from __future__ import annotations

import torch

from ..core import random as rnd

_CHUNK = 1 << 22


def _cover_count(samples, points):
    """Number of boxes [0, p_i) containing each sample."""
    n = points.shape[0]
    step = max(1, _CHUNK // max(n, 1))
    out = torch.empty(samples.shape[0], dtype=torch.int64, device=points.device)
    for s in range(0, samples.shape[0], step):
        blk = samples[s:s + step]
        out[s:s + step] = (blk[:, None, :] < points[None, :, :]).all(-1).sum(1)
    return out


def hypervolume_bounding_cube(key, objs, ref, num_sample):
    points = (objs.float() - ref.to(objs.device)).abs()
    bound = points.amax(0)
    samples = rnd.uniform(key.to(points.device), (num_sample, points.shape[1])) * bound
    inside = (_cover_count(samples, points) > 0).sum()
    return inside / num_sample * bound.prod()


def hypervolume_each_cube(key, objs, ref, num_sample):
    """Sample inside every point's own box; a sample covered by c boxes counts 1/c."""
    points = (objs.float() - ref.to(objs.device)).abs()
    n, m = points.shape
    per = max(1, num_sample // n)
    u = rnd.uniform(key.to(points.device), (n, per, m))
    samples = (u * points[:, None, :]).reshape(-1, m)
    cover = _cover_count(samples, points).clamp_min(1)
    share = (1.0 / cover.to(torch.float32)).reshape(n, per).mean(1)
    return (points.prod(1) * share).sum()


def _slice_hv(pts):
    m = len(pts[0])
    if m == 1:
        return max(p[0] for p in pts)
    pts = sorted(pts, key=lambda p: -p[-1])
    total = 0.0
    for i, p in enumerate(pts):
        nxt = pts[i + 1][-1] if i + 1 < len(pts) else 0.0
        depth = p[-1] - nxt
        if depth > 0:
            total += depth * _slice_hv([q[:-1] for q in pts[: i + 1]])
    return total


def hypervolume_exact(objs, ref):
    """Exact hypervolume by recursive slicing (host side; intended for fronts of a few hundred
    points and up to ~6 objectives)."""
    points = (objs.detach().double().cpu() - ref.detach().double().cpu()).abs().tolist()
    return torch.tensor(_slice_hv([tuple(p) for p in points]), dtype=torch.float32)
