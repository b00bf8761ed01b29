"""NSGA-II (Deb et al. 2002), GDE3 (Kukkonen & Lampinen 2005).
Reference: ``src/evox/algorithms/mo/nsga2.py:22-96``, ``gde3.py:23-98``."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...operators import crossover
from ...operators.selection.non_dominate import NonDominate
from .common import MOAlgorithm, rank_crowding_truncate


class NSGA2(MOAlgorithm):
    """Offspring by SBX + polynomial mutation of the whole population; survivors by
    non-dominated sorting with crowding-distance truncation."""

    def __init__(self, lb, ub, n_objs, pop_size, selection_op=None, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op, selection_op)
        self.survivor_selection = NonDominate(self.pop_size)

    def ask(self, state):
        key, x_key, mut_key = rnd.split(state.key, 3)
        nxt = self._vary(x_key, mut_key, state.population)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, fit = self._merged(state, fitness)
        survivors, survivor_fit = self.survivor_selection(pop, fit)
        return state.replace(population=survivors, fitness=survivor_fit)


class GDE3(MOAlgorithm):
    """DE/rand/1/bin variation with NSGA-II style survivor selection."""

    def __init__(self, lb, ub, n_objs, pop_size, F=0.49, CR=0.97):
        self._init_mo(lb, ub, n_objs, pop_size)
        self.F, self.CR = F, CR
        self.de = crossover.DifferentialEvolve(F, CR)

    def ask(self, state):
        key, k_pick, k_de = rnd.split(state.key, 3)
        idx = rnd.randint(k_pick, (3, self.pop_size), 0, self.pop_size)
        p = state.population[idx]
        nxt = torch.clamp(self.de(k_de, p[0], p[1], p[2]), self.lb, self.ub)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, fit = self._merged(state, fitness)
        keep = rank_crowding_truncate(pop, fit, self.pop_size)
        return state.replace(population=pop[keep], fitness=fit[keep])
