"""Decomposition containers: split the decision vector into equal blocks and run one copy of a
base algorithm per block, all blocks being told the same fitness.
Reference: ``src/evox/algorithms/containers/clustered_algorithm.py:11-160``.

The reference vmaps one algorithm over stacked states, i.e. runs one fused program for all blocks.  Three forms here:

* **fused** – a base algorithm that is separable over decision variables (``dimension_separable``: PSO, CSO) is run
  as ONE instance over the concatenated vector: the same kernels as a single block, num_cluster times wider, no
  per-block launches at all (``_members.widen``);
* **batched** – plain CMA-ES becomes one :class:`~evox_amd.algorithms.so.es_variants.batched_cma.BatchedCMAES`: the
  state carries a leading block axis and every step is one batched product / factorisation for all blocks, so a
  generation costs the launches of one member whatever ``num_cluster`` is (``_members.batch``);
* **member loop** – any other base algorithm, or a list of differently configured members: every block
  owns a member with its own key and state, driven in sequence; each ask/tell is a population-wide kernel sequence,
  and the whole generation is one HIP-graph replay under ``StdWorkflow(graph_step=True)``;
* **distributed** (``distribute=True``, one process per GPU) – the blocks are dealt out over the ranks: rank r evolves
  blocks ``[r·C/W, (r+1)·C/W)``, the sub-populations are all-gathered along the variable axis into the full
  candidates, every rank evaluates its row slice as usual and all are told the same fitness."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.algorithm import Algorithm
from ...core.state import State
from ...core import distributed as dist
from ...core.module import Stateful
from ._members import MemberRecording, as_stack, batch, for_each_member, member_ask, member_tell, widen


class ClusterdAlgorithm(MemberRecording, Algorithm):
    def __init__(self, base_algorithm, dim: int, num_cluster: int, fuse="auto", distribute: bool = False):
        """``fuse``: "auto" (fuse when the base algorithm allows it), True (require it) or False (member loop).
        ``distribute``: deal the blocks out over the ranks of the job (``num_cluster`` divisible by the world size)."""
        super().__init__()
        assert dim % num_cluster == 0
        self.dim, self.num_cluster = dim, num_cluster
        self.subproblem_dim = dim // num_cluster
        self.distribute = bool(distribute)
        local = num_cluster
        if self.distribute:
            world = dist.get_world_size()
            if num_cluster % world:
                raise ValueError(f"num_cluster={num_cluster} is not divisible by the number of ranks {world}")
            local = num_cluster // world
        self.local_clusters = local
        wide = None
        if fuse in ("auto", True) and isinstance(base_algorithm, Stateful) and not isinstance(base_algorithm, (list, tuple)):
            wide = widen(base_algorithm, local)
            if wide is None:
                wide = batch(base_algorithm, local)
        if fuse is True and wide is None:
            raise TypeError(f"{type(base_algorithm).__name__} is neither separable over decision variables nor available in a "
                            "batched form: it cannot be fused")
        self.fused = wide is not None
        if self.fused:
            self.members = as_stack([wide])
        elif self.distribute:
            full = as_stack(base_algorithm, num_cluster)
            rank = dist.get_process_id()
            self.members = as_stack([full[i] for i in range(rank * local, (rank + 1) * local)])
        else:
            self.members = as_stack(base_algorithm, num_cluster)
            assert len(self.members) == num_cluster

    def setup(self, key):
        return State()

    def init(self, key=None, no_state: bool = False):
        # ranks must not evolve identical blocks: every rank derives its own key
        if self.distribute and key is not None and dist.get_world_size() > 1:
            key = rnd.fold_in(key, dist.get_process_id())
        return super().init(key, no_state)

    def _ask(self, state, first):
        pops, state = for_each_member(list(self.members), state, lambda m, s: member_ask(m, s, first))
        pop = pops[0] if len(pops) == 1 else torch.cat(pops, dim=1)
        if self.distribute and dist.get_world_size() > 1:
            # (P, C_local·d) per rank -> (P, C·d): ranks hold consecutive blocks
            w = dist.get_world_size()
            pop = dist.all_gather(pop.T.contiguous(), axis=0, tiled=True).T.contiguous() if w > 1 else pop
        return pop, state

    def _tell(self, state, fitness, first):
        _, state = for_each_member(list(self.members), state, lambda m, s: (None, member_tell(m, s, fitness, first)))
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
