"""Bi-criterion evolution IBEA (Li, Yang & Liu 2016).
Reference: ``src/evox/algorithms/mo/bce_ibea.py:18-335``.

Two populations co-evolve: the *Pareto-criterion* population PC (``population``/``fitness``;
non-dominated solutions thinned by a niche radius) and the *non-Pareto-criterion* population
NPC (``npc``/``npc_obj``; IBEA's ε-indicator selection).  Odd steps let under-explored PC
members produce offspring, even steps evolve NPC; both offspring sets feed both selections.
As with the reference a run needs ``2·generations + 1`` workflow steps.

The reference stores NPC's objectives into ``fitness`` after the odd step and later pairs
``population`` with ``npc_obj`` (``bce_ibea.py:296-330``), so individuals and objective vectors
drift apart; here each population always travels with its own objectives."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.state import State
from ...operators import crossover, selection
from ...operators.selection.non_dominate import non_dominated_sort
from ...ops import moselect as _sel
from ...ops.distance import cdist as _cdist
from .common import MOAlgorithm
from .indicator import ibea_fitness_only, ibea_survivors


def _third_nearest(dist):
    """``(distance to the third nearest other point, distances with the diagonal and NaNs at +inf)``.  Only three values
    per row are wanted: a selection of the three smallest, not a sort of every row."""
    d = dist.clone()
    d.fill_diagonal_(float("inf"))
    d = torch.nan_to_num(d, nan=float("inf"))
    k = min(3, d.shape[1])
    return torch.topk(d, k, dim=1, largest=False, sorted=True).values[:, k - 1], d


def _third_nearest_distance(points):
    """The first result of :func:`_third_nearest` straight from the coordinates (streamed on the GPU: no n × n matrix)."""
    n = points.shape[0]
    if n > 3 and points.is_cuda:
        return _sel.knn_distances(points, 3)[:, 2]
    return _third_nearest(_cdist(points, points))[0]


def exploration(pc_obj, npc_obj, n_nd, n):
    """PC members whose niche (radius scaled by the share of non-dominated solutions) holds at
    most one NPC solution still deserve exploration."""
    f_min, f_max = pc_obj.amin(0), pc_obj.amax(0)
    span = (f_max - f_min).clamp_min(1e-12)
    a, b = (pc_obj - f_min) / span, (npc_obj - f_min) / span
    r0 = _third_nearest_distance(a).mean()
    r = n_nd / n * r0
    return (_cdist(a, b) <= r).sum(1) <= 1


def pc_selection(pc, pc_obj, n):
    """Keep the non-dominated set; when it exceeds ``n`` repeatedly drop the most crowded
    member (largest ``1 - Π_j min(d_ij / r, 1)``).  The crowding products are maintained
    incrementally (log-sum plus a zero counter), O(N) per removal."""
    rank = non_dominated_sort(pc_obj, until=1)             # only membership of the first front matters
    nd = rank == 0
    idx = torch.nonzero(nd)[:, 0]
    n_nd = int(idx.shape[0])
    if n_nd > n:
        obj = pc_obj[idx]
        f_min, f_max = obj.amin(0), obj.amax(0)
        norm = (obj - f_min) / (f_max - f_min).clamp_min(1e-12)
        kth, d = _third_nearest(_cdist(norm, norm))
        r = kth.sum() / n_nd
        big_r = (d / r).clamp_max(1.0)
        idx = idx[_sel.crowd_remove(big_r, n_nd - n)]
    else:
        idx = torch.cat([idx, idx[:1].expand(n - n_nd)])
    return pc[idx], pc_obj[idx], n_nd


def environmental_selection(pop, obj, n, kappa):
    alive = ibea_survivors(obj, kappa, n)
    idx = torch.argsort((~alive).to(torch.int8), stable=True)[:n]
    return pop[idx], obj[idx]


class BCEIBEA(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, kappa=0.05, selection_op=None, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.kappa = kappa
        self.selection = selection.Tournament(n_round=self.pop_size)
        self.crossover_odd = crossover.SimulatedBinary(type=2)

    def setup(self, key):
        f = self._base_fields(key)
        pop, z = f["population"], f["fitness"]
        return State(npc=pop, npc_obj=z, new_pc=pop, new_pc_obj=z, new_npc=pop, new_npc_obj=z, n_nd=0, counter=1, **f)

    def init_tell(self, state, fitness):
        pc, pc_obj, n_nd = pc_selection(state.population, fitness, self.pop_size)
        return state.replace(population=pc, fitness=pc_obj, npc_obj=fitness, new_pc_obj=fitness, new_npc_obj=fitness,
                             n_nd=n_nd)

    def ask(self, state):
        return self._ask_even(state) if state.counter % 2 == 0 else self._ask_odd(state)

    def tell(self, state, fitness):
        return self._tell_even(state, fitness) if state.counter % 2 == 0 else self._tell_odd(state, fitness)

    def _ask_odd(self, state):
        key, k_mate, k_x, k_mut = rnd.split(state.key, 4)
        n = self.pop_size
        s = exploration(state.fitness, state.npc_obj, state.n_nd, n)
        pop = state.population
        if bool(s.any()):
            head = torch.nonzero(s)[0, 0]
            me = torch.where(s, torch.arange(n, device=s.device), head)
            mate = torch.where(s, rnd.randint(k_mate, (n,), 0, n), head)
            off = self.mutation(k_mut, self.crossover_odd(k_x, pop[torch.cat([me, mate])]))
            pop = torch.clamp(off, self.lb, self.ub)
        return pop, state.replace(new_pc=pop, next_generation=pop, key=key)

    def _ask_even(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        fit = -ibea_fitness_only(state.npc_obj, self.kappa)
        parents, _ = self.selection(k_sel, state.npc, fit)
        nxt = self._vary(k_x, k_mut, parents)
        return nxt, state.replace(new_npc=nxt, next_generation=nxt, key=key)

    def _tell_odd(self, state, fitness):
        npc, npc_obj = environmental_selection(torch.cat([state.npc, state.new_pc]),
                                               torch.cat([state.npc_obj, fitness]), self.pop_size, self.kappa)
        return state.replace(npc=npc, npc_obj=npc_obj, new_pc_obj=fitness, counter=state.counter + 1)

    def _tell_even(self, state, fitness):
        npc, npc_obj = environmental_selection(torch.cat([state.npc, state.new_npc]),
                                               torch.cat([state.npc_obj, fitness]), self.pop_size, self.kappa)
        pc, pc_obj, n_nd = pc_selection(torch.cat([state.population, state.new_npc, state.new_pc]),
                                        torch.cat([state.fitness, fitness, state.new_pc_obj]), self.pop_size)
        return state.replace(population=pc, fitness=pc_obj, n_nd=n_nd, new_npc_obj=fitness, npc=npc, npc_obj=npc_obj,
                             counter=state.counter + 1)
