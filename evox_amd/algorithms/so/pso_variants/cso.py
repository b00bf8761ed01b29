"""Competitive swarm optimiser (Cheng & Jin 2015).  Reference: ``pso_variants/cso.py:24-100``:
random pairwise competitions; each loser learns from its winner and from the swarm centre and
only the N/2 updated losers are re-evaluated."""
from __future__ import annotations

from typing import Optional

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.struct import dataclass, pytree_field
from ....ops import swarm as _swarm


@dataclass
class CSOState:
    population: torch.Tensor
    fitness: torch.Tensor
    velocity: torch.Tensor
    students: torch.Tensor
    key: torch.Tensor


@dataclass
class CSO(Algorithm):
    dimension_separable = True          # per-variable update given the fitness: containers may fuse blocks

    lb: torch.Tensor
    ub: torch.Tensor
    pop_size: int = pytree_field(static=True)
    phi: float = 0.0
    mean: Optional[torch.Tensor] = None
    stdev: Optional[torch.Tensor] = None
    dim: int = pytree_field(static=True, init=False, default=0)

    def __post_init__(self):
        object.__setattr__(self, "lb", torch.as_tensor(self.lb, dtype=torch.float32))
        object.__setattr__(self, "ub", torch.as_tensor(self.ub, dtype=torch.float32))
        self.set_frozen_attr("dim", self.lb.shape[-1])

    def setup(self, key):
        dev = self.lb.device
        state_key, init_key = rnd.split(key.to(dev))
        shape = (self.pop_size, self.dim)
        if self.mean is not None and self.stdev is not None:
            population = torch.clamp(self.stdev * rnd.normal(init_key, shape) + self.mean, self.lb, self.ub)
        else:
            population = rnd.uniform(init_key, shape) * (self.ub - self.lb) + self.lb
        return CSOState(population=population, fitness=torch.full((self.pop_size,), float("inf"), device=dev),
                        velocity=torch.zeros(shape, device=dev),
                        students=torch.zeros(self.pop_size // 2, dtype=torch.int64, device=dev), key=state_key)

    def init_ask(self, state):
        return state.population, state

    def init_tell(self, state, fitness):
        return state.replace(fitness=fitness)

    def ask(self, state):
        key, k_pair, k_r = rnd.split(state.key, 3)
        half = self.pop_size // 2
        perm = rnd.permutation(k_pair, self.pop_size)[: 2 * half].reshape(2, half)
        if (_swarm.on_gpu(state.population) and state.population.dtype == torch.float32 and half > 0
                and state.population.dim() == 2 and state.fitness.dim() == 1):
            cand, population, velocity, students = _swarm.cso_step(k_r, state.population, state.velocity, state.fitness,
                                                                   perm, self.phi, self.lb, self.ub)
            return cand, state.replace(population=population, velocity=velocity, students=students, key=key)
        first_wins = state.fitness[perm[0]] < state.fitness[perm[1]]
        teachers = torch.where(first_wins, perm[0], perm[1])
        students = torch.where(first_wins, perm[1], perm[0])
        lam = rnd.uniform(k_r, (3, half, self.dim))
        center = state.population.mean(0)
        xs = state.population[students]
        vel = lam[0] * state.velocity[students] + lam[1] * (state.population[teachers] - xs) \
            + self.phi * lam[2] * (center - xs)
        cand = torch.clamp(xs + vel, self.lb, self.ub)
        population = state.population.index_copy(0, students, cand)
        velocity = state.velocity.index_copy(0, students, vel)
        return cand, state.replace(population=population, velocity=velocity, students=students, key=key)

    def tell(self, state, fitness):
        return state.replace(fitness=state.fitness.index_copy(0, state.students, fitness))
