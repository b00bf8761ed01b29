"""Decomposition containers: split the decision vector into equal blocks and run one copy of a
base algorithm per block, all blocks being told the same fitness.
Reference: ``src/evox/algorithms/containers/clustered_algorithm.py:11-160``.

The reference vmaps one algorithm over stacked states.  Here every block owns a member
algorithm (own key, own state) driven in sequence – each member's ask/tell is already a
population-wide kernel sequence, so blocks cost launches, not efficiency; members may also
differ in hyper-parameters (pass a list)."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.algorithm import Algorithm
from ...core.state import State
from ._members import as_stack, member_ask, member_tell


class ClusterdAlgorithm(Algorithm):
    def __init__(self, base_algorithm, dim: int, num_cluster: int):
        super().__init__()
        assert dim % num_cluster == 0
        self.dim, self.num_cluster = dim, num_cluster
        self.subproblem_dim = dim // num_cluster
        self.members = as_stack(base_algorithm, num_cluster)
        assert len(self.members) == num_cluster

    def _ask(self, state, first):
        pops = []
        for m in self.members:
            pop, state = member_ask(m, state, first)
            pops.append(pop)
        return torch.cat(pops, dim=1), state

    def _tell(self, state, fitness, first):
        for m in self.members:
            state = member_tell(m, state, fitness, first)
        return state

    def init_ask(self, state):
        return self._ask(state, True)

    def init_tell(self, state, fitness):
        return self._tell(state, fitness, True)

    def ask(self, state):
        return self._ask(state, False)

    def tell(self, state, fitness):
        return self._tell(state, fitness, False)


class RandomMaskAlgorithm(Algorithm):
    """Like :class:`ClusterdAlgorithm`, but only ``num_cluster - num_mask`` randomly chosen blocks
    evolve at a time; the others keep their last sub-populations.  The active set is redrawn
    (without replacement) every ``change_every`` generations."""

    def __init__(self, base_algorithm, dim: int, num_cluster: int, num_mask: int = 1, change_every: int = 1,
                 pop_size=None):
        super().__init__()
        assert dim % num_cluster == 0
        assert 0 < num_mask < num_cluster
        self.dim, self.num_cluster, self.num_mask = dim, num_cluster, num_mask
        self.num_valid = num_cluster - num_mask
        self.change_every = change_every
        self.subproblem_dim = dim // num_cluster
        self.pop_size = base_algorithm.pop_size if pop_size is None else pop_size
        self.members = as_stack(base_algorithm, num_cluster)

    def setup(self, key):
        return State(key=key, sub_pops=None, active=tuple(range(self.num_valid)), count=0)

    def init_ask(self, state):
        pops = []
        for m in self.members:
            pop, state = member_ask(m, state, True)
            pops.append(pop)
        return torch.cat(pops, dim=1), state.replace(sub_pops=pops)

    def init_tell(self, state, fitness):
        for m in self.members:
            state = member_tell(m, state, fitness, True)
        return state

    def ask(self, state):
        if state.count > 0 and state.count % self.change_every == 0:
            key, sub = rnd.split(state.key)
            active = tuple(sorted(rnd.permutation(sub, self.num_cluster)[: self.num_valid].tolist()))
            state = state.replace(key=key, active=active)
        sub_pops = list(state.sub_pops)
        rows = None
        for i in state.active:
            pop, state = member_ask(self.members[i], state, False)
            sub_pops[i] = pop
            rows = pop.shape[0]
        # a base algorithm may ask for fewer candidates than it holds (CSO evaluates only the
        # losers' moves): frozen blocks contribute their first ``rows`` stored rows
        blocks = [p[:rows] if p.shape[0] >= rows else p.repeat(-(-rows // p.shape[0]), 1)[:rows] for p in sub_pops]
        return torch.cat(blocks, dim=1), state.replace(sub_pops=sub_pops)

    def tell(self, state, fitness):
        for i in state.active:
            state = member_tell(self.members[i], state, fitness, False)
        return state.replace(count=state.count + 1)
