"""Indicator- and strength-based MOEAs: IBEA, SPEA2, HypE.
Reference: ``src/evox/algorithms/mo/ibea.py:20-132``, ``spea2.py:20-158``, ``hype.py:20-147``.

IBEA and SPEA2 shrink the merged population by removing one individual at a time, each removal
changing the scores of the others; these loops are inherently sequential in the number of
removals and run on the device through ``evox_amd.ops.moselect`` (one persistent launch on a
GPU, a tight host loop otherwise)."""
from __future__ import annotations

import math

import torch

from ...ops._util import pick
from ... import _native
from ...core import random as rnd
from ...core.state import State
from ...operators import selection
from ...operators.selection.non_dominate import non_dominated_sort
from ...ops import moselect as _sel
from ...ops._util import check as _check
from ...ops._util import hip as _hip
from ...ops._util import stream_of as _stream_of
from ...ops.distance import cdist as _cdist
from ...utils.common import cal_max, dominate_relation
from .common import MOAlgorithm


def _normalised(pop_obj):
    lo, hi = pop_obj.min(), pop_obj.max()
    return (pop_obj - lo) / (hi - lo)


def ibea_fitness(pop_obj, kappa):
    """``(fitness, indicator matrix, C)`` with the dense n × n ε-indicator matrix."""
    obj = _normalised(pop_obj)
    I = cal_max(obj, obj)
    C = I.abs().amax(0)
    fit = (-torch.exp(-I / C[None, :] / kappa)).sum(0) + 1
    return fit, I, C


def ibea_fitness_only(pop_obj, kappa):
    """The fitness alone; on a GPU without forming the indicator matrix."""
    if _sel.ibea_streaming_supported(pop_obj):
        return _sel.ibea_fitness(_normalised(pop_obj), kappa)[0]
    return ibea_fitness(pop_obj, kappa)[0]


def ibea_survivors(pop_obj, kappa, n_keep):
    """Boolean mask of the ``n_keep`` members IBEA's environmental selection retains."""
    n = pop_obj.shape[0]
    if _sel.ibea_streaming_supported(pop_obj) and _sel.ibea_remove_streaming_fits(n, pop_obj.shape[1]):
        obj = _normalised(pop_obj)
        fit, C = _sel.ibea_fitness(obj, kappa)
        return _sel.ibea_remove_streaming(obj, fit, C, kappa, n - n_keep)
    fit, I, C = ibea_fitness(pop_obj, kappa)
    return _sel.ibea_remove(fit, I, C, kappa, n - n_keep)


class IBEA(MOAlgorithm):
    """Indicator-based EA with the additive ε-indicator; ``kappa`` scales the fitness."""

    def __init__(self, lb, ub, n_objs, pop_size, kappa=0.05, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.kappa = kappa
        self.selection = selection.Tournament(n_round=self.pop_size)

    def ask(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        fit = ibea_fitness_only(state.fitness, self.kappa)
        parents, _ = self.selection(k_sel, state.population, -fit)
        nxt = self._vary(k_x, k_mut, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, obj = self._merged(state, fitness)
        alive = ibea_survivors(obj, self.kappa, self.pop_size)
        idx = torch.argsort((~alive).to(torch.int8), stable=True)[: self.pop_size]
        return state.replace(population=pop[idx], fitness=obj[idx])


def spea2_fitness(obj):
    """Raw fitness (sum of the strengths of one's dominators) + k-th-nearest-neighbour density."""
    if _sel.spea2_fitness_supported(obj) and obj.shape[0] >= 2:
        return _sel.spea2_fitness(obj)
    return spea2_fitness_reference(obj)


def spea2_fitness_reference(obj):
    """``spea2_fitness`` with dense n × n dominance and distance matrices."""
    n = obj.shape[0]
    dom = dominate_relation(obj, obj)
    strength = dom.sum(1).to(torch.float32)
    raw = (dom.to(torch.float32) * strength[:, None]).sum(0)
    dis = _cdist(obj, obj)
    dis = dis.masked_fill(torch.eye(n, dtype=torch.bool, device=obj.device), float("inf"))
    k = int(math.floor(math.sqrt(6)))
    kth = torch.topk(dis, k, dim=1, largest=False).values[:, k - 1]
    return 1.0 / (kth + 2) + raw


class SPEA2(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.selection = selection.Tournament(n_round=self.pop_size)

    def ask(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        parents, _ = self.selection(k_sel, state.population, spea2_fitness(state.fitness))
        nxt = self._vary(k_x, k_mut, parents)
        # the reference forgets to store the advanced key here (``spea2.py:128``)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, obj = self._merged(state, fitness)
        sig = spea2_fitness(obj)
        nondominated = sig < 1
        n_valid = int(nondominated.sum())
        if n_valid <= self.pop_size:
            idx = torch.argsort(sig, stable=True)[: self.pop_size]
        else:
            keep = _sel.spea2_truncate(obj, nondominated, n_valid - self.pop_size)
            idx = torch.argsort((~keep).to(torch.int8), stable=True)[: self.pop_size]
        return state.replace(population=pop[idx], fitness=obj[idx])


_native.register_hip_api({"evox_hip_hv_credit": (_native.I, [_native.P, _native.I, _native.P, _native.I, _native.I, _native.P,
                                                            _native.P, _native.P, _native.I, _native.P])})


def _hv_kernel_enabled() -> bool:
    import os

    return os.environ.get("EVOX_AMD_HV", "hip") == "hip"


def cal_hv(points, ref, k, n_sample, key, tight: bool = False):
    """HypE's Monte-Carlo estimate of each point's hypervolume contribution when ``k`` points
    are to be removed: a sample dominated by j points credits each of them α_j.

    ``tight``: sample the box between the points' minimum and ``min(ref, 1.1 · their maximum)`` instead of the box up
    to ``ref``.  The reference point is fixed at 1.2 × the largest objective of the initial population
    (``hype.py:107-110``); once the population has converged most of that box lies far behind the front, where a sample
    is dominated by (almost) every point and tells them nothing apart — with the default 10 000 samples and a few
    thousand points that leaves a few hundred informative samples.  The slab between 1.1 · max and ``ref`` adds the same
    credit to every point whose other coordinates it covers, so dropping it changes the ranking only at the extremes."""
    n, m = points.shape
    dev = points.device
    if tight:
        ref = torch.minimum(ref, points.amax(0) * 1.1 + 1e-12)
    i = torch.arange(1, n, device=dev, dtype=torch.float32)
    # α_j = Π_{i<j} (k−i)/(n−i) / j vanishes for j > k; clamping the factors keeps a parallel scan from
    # multiplying that zero with an overflowed tail (the factors beyond k grow up to n−1−k)
    alpha = torch.cumprod(torch.cat([torch.ones(1, device=dev), (k - i).clamp_min(0) / (n - i)]), 0) / torch.arange(
        1, n + 1, device=dev, dtype=torch.float32)
    f_min = points.amin(0)
    samples = f_min + rnd.uniform(key, (n_sample, m)) * (ref - f_min)
    if points.is_cuda and points.dtype == torch.float32 and 2 <= m <= 8 and _hv_kernel_enabled():
        # two scans on the GPU instead of (n × S) boolean blocks (csrc/hvmc.hip)
        ysplit = 8
        cnt = torch.zeros(n_sample, dtype=torch.int32, device=dev)
        partial = torch.empty((ysplit, n), dtype=torch.float32, device=dev)
        _check(_hip().evox_hip_hv_credit(points.contiguous().data_ptr(), n, samples.contiguous().data_ptr(), n_sample, m,
                                         cnt.data_ptr(), alpha.contiguous().data_ptr(), partial.data_ptr(), ysplit,
                                         _stream_of(points)), "hv_credit")
        return partial.sum(0) * (ref - f_min).prod() / n_sample
    credit = torch.zeros(n, device=dev)
    step = max(1, (1 << 22) // max(n, 1))
    for s in range(0, n_sample, step):
        blk = samples[s:s + step]
        dominated = (points[:, None, :] <= blk[None, :, :]).all(-1)                # (n, S)
        cnt = dominated.sum(0)
        a = torch.where(cnt > 0, alpha[(cnt - 1).clamp_min(0)], torch.zeros_like(cnt, dtype=torch.float32))
        credit += (dominated.to(torch.float32) * a[None, :]).sum(1)
    return credit * (ref - f_min).prod() / n_sample


class HypE(MOAlgorithm):
    """Environmental selection: whole fronts, then the cut front by estimated hypervolume contribution (the reference
    applies the estimate to the last front of the merged population instead, ``hype.py:130-140``, which never decides a
    cut).  ``tight_sampling`` (default) confines the Monte-Carlo samples to the region where they separate points, see
    :func:`cal_hv`; ``False`` samples the reference's fixed box."""

    def __init__(self, lb, ub, n_objs, pop_size, n_sample=10000, mutation_op=None, crossover_op=None,
                 tight_sampling: bool = True):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.n_sample = n_sample
        self.tight_sampling = bool(tight_sampling)
        self.selection = selection.Tournament(n_round=self.pop_size, multi_objective=True)

    def setup(self, key):
        return State(ref_point=torch.zeros(self.n_objs, device=self.device), **self._base_fields(key))

    def init_tell(self, state, fitness):
        ref = torch.zeros(self.n_objs, device=self.device) + fitness.max() * 1.2
        return state.replace(fitness=fitness, ref_point=ref)

    def ask(self, state):
        key, k_hv, k_sel, k_x, k_mut = rnd.split(state.key, 5)
        hv = cal_hv(state.fitness, state.ref_point, self.pop_size, self.n_sample, k_hv, self.tight_sampling)
        parents, _ = self.selection(k_sel, state.population, -hv)
        nxt = self._vary(k_x, k_mut, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, obj = self._merged(state, fitness)
        n = obj.shape[0]
        key, k_hv = rnd.split(state.key)
        rank = non_dominated_sort(obj, until=self.pop_size)
        order = torch.argsort(rank, stable=True)
        cut_rank = pick(rank, order[self.pop_size - 1])
        mask = rank == cut_rank
        # k = the number of points that will be removed, as in the HypE paper (the reference passes k = n, ``hype.py:136``,
        # which weights a sample dominated by j points with 1/j whatever j is)
        hv = cal_hv(obj, state.ref_point, n - self.pop_size, self.n_sample, k_hv, self.tight_sampling)
        tie = torch.where(mask, -hv, torch.zeros_like(hv))
        idx = torch.argsort(tie, stable=True)
        idx = idx[torch.argsort(rank[idx], stable=True)][: self.pop_size]
        return state.replace(population=pop[idx], fitness=obj[idx], key=key)
