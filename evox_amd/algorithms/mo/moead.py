"""MOEA/D (Zhang & Li 2007) and MOEA/D-DRA (Zhang, Liu & Li 2009).
Reference: ``src/evox/algorithms/mo/moead.py:18-133``, ``moeaddra.py:23-203``.

MOEA/D's reference implementation walks the offspring one by one and lets each replace the
neighbours it improves.  With the ideal point fixed for the generation this is a running
minimum per sub-problem, so the survivor of sub-problem j is simply the best candidate among
its current solution and every offspring whose neighbourhood contains j (earliest offspring on
ties): one scatter-min over the (offspring × neighbour) table instead of a sequential scan."""
from __future__ import annotations

import math

import torch

from ...core import random as rnd
from ...core.state import State
from ...operators import crossover, selection
from ...operators.sampling import LatinHypercubeSampling, UniformSampling
from ...ops import moselect as _sel
from ...ops.distance import cdist as _cdist
from ...utils.common import AggregationFunction
from .common import MOAlgorithm


def _scatter_best(values, target, n_sub, current):
    """For each sub-problem j: (best value, row index of the entry achieving it or -1 when the
    current value is not beaten).  ``values`` (N, T) with targets ``target`` (N, T).

    On the GPU one pass with a 64-bit atomic minimum on (value, row) per improving entry (``ops.moselect.scatter_best``);
    the tensor formulation below is two atomic scatter passes over all N·T entries."""
    if values.is_cuda and values.dtype == torch.float32 and values.dim() == 2 and current.dtype == torch.float32:
        return _sel.scatter_best(values, target, n_sub, current)
    flat_v, flat_t = values.reshape(-1), target.reshape(-1)
    best = current.clone().scatter_reduce(0, flat_t, flat_v, reduce="amin", include_self=True)
    improved = flat_v < current[flat_t]
    hit = improved & (flat_v == best[flat_t])
    row = torch.arange(values.shape[0], device=values.device)[:, None].expand_as(values).reshape(-1)
    big = values.shape[0]
    winner = torch.full((n_sub,), big, dtype=torch.int64, device=values.device).scatter_reduce(
        0, flat_t, torch.where(hit, row, torch.full_like(row, big)), reduce="amin", include_self=True)
    return best, torch.where(winner < big, winner, torch.full_like(winner, -1))


def two_distinct(key, n: int, t: int):
    """For each of ``n`` rows two different positions in ``[0, t)``, uniform over the ordered pairs — the first two
    entries of a random permutation of the row (what the reference draws, ``moead.py:88-92``) without sorting ``t``
    random keys per row."""
    k1, k2 = rnd.split(key)
    first = rnd.randint(k1, (n,), 0, t)
    second = rnd.randint(k2, (n,), 0, max(t - 1, 1))
    second = second + (second >= first).to(second.dtype) if t > 1 else second
    return first, second


class MOEAD(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, func_name="pbi", mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op,
                      crossover_op if crossover_op is not None else crossover.SimulatedBinary(type=2))
        self.func_name = func_name
        self.aggregate_func = AggregationFunction(func_name)
        w, n = UniformSampling(self.pop_size, self.n_objs)()
        self._w = w.to(self.device)
        self.pop_size = n
        self.n_neighbor = int(math.ceil(n / 10))

    def setup(self, key):
        f = self._base_fields(key)
        nb = torch.argsort(_cdist(self._w, self._w), dim=1, stable=True)[:, : self.n_neighbor]
        return State(weight_vector=self._w, neighbors=nb, z=torch.zeros(self.n_objs, device=self.device), **f)

    def init_tell(self, state, fitness):
        return state.replace(fitness=fitness, z=fitness.amin(0))

    def ask(self, state):
        key, k_perm, k_x, k_mut = rnd.split(state.key, 4)
        n, t = state.neighbors.shape
        first, second = two_distinct(k_perm, n, t)
        parent = torch.gather(state.neighbors, 1, torch.stack([first, second], dim=1))
        mates = torch.cat([state.population[parent[:, 0]], state.population[parent[:, 1]]], dim=0)
        nxt = self._vary(k_x, k_mut, mates)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        w, nb = state.weight_vector, state.neighbors
        z = torch.minimum(state.z, fitness.amin(0))
        z_max = state.fitness.amax(0)
        n, t = nb.shape
        g_old = self.aggregate_func(state.fitness, w, z, z_max)                                   # (n,)
        g_new = self.aggregate_func(fitness[:, None, :].expand(n, t, -1).reshape(n * t, -1),
                                    w[nb].reshape(n * t, -1), z, z_max).reshape(n, t)
        _, winner = _scatter_best(g_new, nb, n, g_old)
        take = winner >= 0
        src = winner.clamp_min(0)
        population = torch.where(take[:, None], state.next_generation[src], state.population)
        pop_obj = torch.where(take[:, None], fitness[src], state.fitness)
        return state.replace(population=population, fitness=pop_obj, z=z)


class MOEADDRA(MOAlgorithm):
    """MOEA/D with dynamic resource allocation: sub-problems are picked for variation by a
    10-tournament on their utility π, offspring come from DE/rand/1 + polynomial mutation, and
    each offspring may replace at most ``nr`` solutions.

    Replacement is *synchronous*: every offspring tests its candidate sub-problems against the
    population as it stood at the start of the generation (with the generation's final ideal
    point), keeps its first ``nr`` improvable targets, and conflicts are resolved in favour of
    the best offspring.  The reference processes offspring strictly one after another
    (``moeaddra.py:151-172``); the two differ only when several offspring of one generation
    compete for the same sub-problem."""

    def __init__(self, lb, ub, n_objs, pop_size, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op,
                      crossover_op if crossover_op is not None else crossover.DifferentialEvolve())
        self.T = int(math.ceil(self.pop_size / 10))
        self.nr = int(math.ceil(self.pop_size / 100))
        self.i_size = int(math.floor(self.pop_size / 5))
        self.selection = selection.Tournament(n_round=self.pop_size)
        self.sample = LatinHypercubeSampling(self.pop_size, self.n_objs)
        self.aggregate_func = AggregationFunction("tchebycheff")

    def setup(self, key):
        f = self._base_fields(key)
        key, k_w = rnd.split(f["key"])
        f["key"] = key
        w = self.sample(k_w)
        B = torch.argsort(_cdist(w, w), dim=1, stable=True)[:, : self.T]
        n, dev = self.pop_size, self.device
        return State(weight_vector=w, B=B, Z=torch.zeros(self.n_objs, device=dev), pi=torch.ones(n, device=dev),
                     old_obj=torch.zeros(n, device=dev), choosed_p=torch.zeros((n, self.T), dtype=torch.int64, device=dev),
                     I_all=torch.zeros(n, dtype=torch.int64, device=dev), gen=0, **f)

    def init_tell(self, state, fitness):
        Z = fitness.amin(0)
        old = ((fitness - Z).abs() * state.weight_vector).amax(1)
        return state.replace(fitness=fitness, Z=Z, old_obj=old)

    def ask(self, state):
        key, k_perm, k_r, k_rp, k_sel, k_x, k_mut = rnd.split(state.key, 7)
        n, T = self.pop_size, self.T
        parent = torch.gather(state.B, 1, torch.argsort(rnd.bits(k_perm, (n, T)), dim=1))
        local = rnd.uniform(k_r, (n, 1)) < 0.9
        anywhere = rnd.randint(k_rp, (n, T), 0, n)
        _, picked = self.selection(k_sel, state.population, -state.pi)
        # the sub-problems on the boundary of the simplex are always included (5 copies)
        w = state.weight_vector
        on_edge = (w < 1e-3).sum(1) == (self.n_objs - 1)
        edge_idx = torch.argsort((~on_edge).to(torch.int8), stable=True)[: self.i_size]
        edge_idx = torch.where(on_edge[edge_idx], edge_idx, torch.zeros_like(edge_idx))
        forced = edge_idx.repeat(5)[:n]
        forced = torch.nn.functional.pad(forced, (0, n - forced.shape[0]))
        I_all = torch.where(forced != 0, forced, picked)
        choosed = torch.where(local, parent[I_all], anywhere)
        pop = state.population
        nxt = self.mutation(k_mut, self.crossover(k_x, pop[I_all], pop[choosed[:, 0]], pop[choosed[:, 1]]))
        nxt = torch.clamp(nxt, self.lb, self.ub)
        return nxt, state.replace(next_generation=nxt, choosed_p=choosed, key=key, I_all=I_all)

    # recorded generations: the counter only decides which generations refresh the utilities (every tenth); it stays
    # the generation count (advanced on the host at replay), the branch is named by ``graph_variant``
    def graph_host_counters(self):
        return ("gen",)

    def graph_variant(self, state):
        return (state.gen + 1) % 10 == 0

    def tell(self, state, fitness):
        gen = state.gen + 1
        w, p = state.weight_vector, state.choosed_p
        n, T = p.shape
        Z = torch.minimum(state.Z, fitness.amin(0))
        g_old = self.aggregate_func(state.fitness, w, Z)
        g_new = self.aggregate_func(fitness[:, None, :].expand(n, T, -1).reshape(n * T, -1), w[p].reshape(n * T, -1),
                                    Z).reshape(n, T)
        can = g_old[p] >= g_new
        allowed = can & (torch.cumsum(can.to(torch.int64), dim=1) <= self.nr)
        vals = torch.where(allowed, g_new, torch.full_like(g_new, float("inf")))
        # >= semantics: an equal value also replaces, so compare against a slightly raised bar
        _, winner = _scatter_best(vals, p, n, torch.nextafter(g_old, torch.full_like(g_old, float("inf"))))
        take = winner >= 0
        src = winner.clamp_min(0)
        population = torch.where(take[:, None], state.next_generation[src], state.population)
        pop_obj = torch.where(take[:, None], fitness[src], state.fitness)
        pi, old_obj = state.pi, state.old_obj
        if gen % 10 == 0:
            new_obj = ((pop_obj - Z).abs() * w).amax(1)
            delta = (old_obj - new_obj) / old_obj
            pi = torch.where(delta < 0.001, pi * (0.95 + 0.05 * delta / 0.001), torch.ones_like(pi))
            old_obj = new_obj
        return state.replace(population=population, fitness=pop_obj, Z=Z, gen=gen, pi=pi, old_obj=old_obj)
