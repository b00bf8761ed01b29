"""Shared pieces of the multi-objective algorithms: bounded random initialisation, the
evaluate-the-initial-population first step, SBX + polynomial-mutation variation, and
rank/crowding truncation."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.algorithm import Algorithm
from ...core.state import State
from ...operators import crossover, mutation
from ...operators.selection.non_dominate import crowding_distance, non_dominated_sort
from ...ops import nds as _nds
from ...ops import variation as _variation


class MOAlgorithm(Algorithm):
    def _init_mo(self, lb, ub, n_objs, pop_size, mutation_op=None, crossover_op=None, selection_op=None):
        super().__init__()
        self.lb = torch.as_tensor(lb, dtype=torch.float32)
        self.ub = torch.as_tensor(ub, dtype=torch.float32).to(self.lb.device)
        self.n_objs = n_objs
        self.dim = self.lb.shape[0]
        self.pop_size = int(pop_size)
        self.selection = selection_op
        self.mutation = mutation_op if mutation_op is not None else mutation.Polynomial((self.lb, self.ub))
        self.crossover = crossover_op if crossover_op is not None else crossover.SimulatedBinary()

    @property
    def device(self):
        return self.lb.device

    def _random_population(self, key, n=None):
        n = self.pop_size if n is None else n
        return rnd.uniform(key, (n, self.dim)) * (self.ub - self.lb) + self.lb

    def _base_fields(self, key):
        key, sub = rnd.split(key.to(self.device))
        population = self._random_population(sub)
        return dict(population=population, fitness=torch.zeros((self.pop_size, self.n_objs), device=self.device),
                    next_generation=population, key=key)

    def setup(self, key):
        return State(**self._base_fields(key))

    def init_ask(self, state):
        return state.population, state

    def init_tell(self, state, fitness):
        return state.replace(fitness=fitness)

    def _vary(self, x_key, mut_key, parents):
        cx, mt = self.crossover, self.mutation
        if (type(cx) is crossover.SimulatedBinary and cx.type == 1 and type(mt) is mutation.Polynomial
                and mt.boundary[0] is self.lb and mt.boundary[1] is self.ub
                and _variation.sbx_pm_supported(parents, self.lb, self.ub)):
            # the default operator pair on the GPU: one fused launch, same random draws
            return _variation.sbx_pm(x_key, mut_key, parents, self.lb, self.ub, cx.pro_c, cx.dis_c, mt.pro_m, mt.dis_m)
        off = mt(mut_key, cx(x_key, parents))
        return torch.clamp(off, self.lb, self.ub)

    def _merged(self, state, fitness):
        return (torch.cat([state.population, state.next_generation], dim=0),
                torch.cat([state.fitness, fitness], dim=0))


def rank_crowding_truncate(pop, fit, k):
    """Indices of the ``k`` survivors by (front index, then larger crowding distance in the
    front that straddles the cut)."""
    return _nds.rank_crowding_select(fit, k)
