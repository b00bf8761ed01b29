"""DE (Storn & Price 1997) and opposition-based DE.  Reference: ``de_variants/de.py:10-166``,
``de_variants/ode.py:17-192``."""
from __future__ import annotations

import torch

from ....core import random as rnd
from .base import BEST, BIN, RAND, DEBase


class DE(DEBase):
    """DE/{rand,best}/k/bin with a sliding window of ``batch_size`` trial vectors per step."""

    def __init__(self, lb, ub, pop_size, base_vector="rand", num_difference_vectors=1, differential_weight=0.5,
                 cross_probability=0.9, batch_size=100, replace=False, mean=None, stdvar=None):
        if pop_size < 4:
            raise ValueError("DE needs at least 4 individuals")
        if not 0 < cross_probability <= 1:
            raise ValueError("cross_probability must be in (0, 1]")
        if base_vector not in ("rand", "best"):
            raise ValueError("base_vector must be 'rand' or 'best'")
        self._init_common(lb, ub, pop_size, min(batch_size, pop_size))
        self.base_vector = base_vector
        self.num_difference_vectors = num_difference_vectors
        self.differential_weight = differential_weight
        self.cross_probability = cross_probability
        self.replace = replace
        self.mean, self.stdvar = mean, stdvar

    def setup(self, key):
        return self._base_state(key)

    def ask(self, state):
        key, k_var = rnd.split(state.key)
        idx = self._window(state)
        code = BEST if self.base_vector == "best" else RAND
        trial = self._strategy_trials(k_var, state, idx, code, code, self.num_difference_vectors, BIN,
                                      self.differential_weight, self.cross_probability,
                                      2 * self.num_difference_vectors + 1, self.replace)
        return trial, state.replace(trial_vectors=trial, key=key)


class ODE(DE):
    """DE whose odd generations evaluate the opposite points ``lb + ub − x`` of the base
    vectors (opposition-based learning), even generations being ordinary DE steps."""

    def setup(self, key):
        return super().setup(key).replace(counter=0)

    def graph_host_counters(self):
        return tuple(super().graph_host_counters()) + ("counter",)

    def graph_variant(self, state):
        return state.counter % 2            # ordinary and opposition generations alternate

    def ask(self, state):
        if state.counter % 2 == 0:
            return super().ask(state)
        key, k_var = rnd.split(state.key)
        idx = self._window(state)
        if self.base_vector == "best":
            base = state.population[state.best_index].expand(idx.shape[0], -1)
        else:
            B = self.blocks
            who = idx.repeat_interleave(B) if B > 1 else idx
            pick = rnd.choice_distinct(k_var, self.pop_size, who.shape[0], 1, exclude=who)[:, 0]
            if B > 1:                                       # a blocked instance: one random individual per (row, block)
                blk = torch.arange(B, device=pick.device).repeat(idx.shape[0])
                base = state.population.reshape(self.pop_size * B, -1)[pick * B + blk].reshape(idx.shape[0], self.dim)
            else:
                base = state.population[pick]
        trial = self.ub + self.lb - base
        return trial, state.replace(trial_vectors=trial, key=key)

    def tell(self, state, trial_fitness):
        return super().tell(state, trial_fitness).replace(counter=state.counter + 1)
