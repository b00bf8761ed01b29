"""Cooperative co-evolution: sub-populations over blocks of the decision vector are evaluated in
the context of the best-known full solution.
Reference: ``src/evox/algorithms/containers/coevolution.py:11-274``.

``VectorizedCoevolution`` evolves all sub-populations in the same generation (the workflow
evaluates ``num_subpops × pop_size`` context vectors at once); ``Coevolution`` evolves them
round-robin, one per generation.

When the members of a ``VectorizedCoevolution`` are identically configured plain CMA-ES, DE, JaDE, SHADE or SepCMAES
instances they are replaced by ONE batched / blocked instance (``_members.batch``) told one fitness per sub-population
and individual: the generation then costs the launches of one member instead of ``num_subpops`` times that
(``fuse=False`` keeps the member loop)."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.algorithm import Algorithm
from ...core.state import State
from ._members import MemberRecording, as_stack, batch, member_ask, member_tell, same_configuration


class _CoevolutionBase(Algorithm):
    def __init__(self, base_algorithms, dim: int, num_subpops: int, random_subpop: bool = False, dtype=torch.float32,
                 fuse="auto"):
        super().__init__()
        assert dim % num_subpops == 0
        self.base_algorithms = as_stack(base_algorithms, num_subpops, stacked=True)
        assert len(self.base_algorithms) == num_subpops
        one = None
        if fuse in ("auto", True) and self._can_batch and same_configuration(list(self.base_algorithms)):
            one = batch(self.base_algorithms[0], num_subpops, concat=False)
        if fuse is True and one is None:
            raise TypeError("the members have no common batched form")
        if one is not None:
            # the instance is a child of this module once, as the only member of the stack (its state, host counters and
            # recording variants are found under that path)
            self.base_algorithms = as_stack([one])
        self._one_instance = one is not None
        self.dim, self.num_subpops, self.random_subpop, self.dtype = dim, num_subpops, random_subpop, dtype
        self.sub_dim = dim // num_subpops

    _can_batch = False           # round-robin co-evolution steps one member per generation: nothing to batch

    @property
    def batched(self):
        """The one batched / blocked instance that stands for all members, or None (member loop)."""
        return self.base_algorithms[0] if self._one_instance else None

    def _base_state(self, key, **extra):
        permutation = rnd.permutation(key, self.dim) if self.random_subpop else None
        return State(coop_pops=None, best_dec=None, best_fit=None, permutation=permutation, **extra)

    def _scatter(self, state, pop):
        """Block order → problem order (column ``permutation[j]`` receives block column j)."""
        if not self.random_subpop:
            return pop
        out = torch.empty_like(pop)
        out[:, state.permutation.to(pop.device)] = pop
        return out

    def _gather(self, state, vec):
        return vec[..., state.permutation.to(vec.device)] if self.random_subpop else vec

    # One instance standing for all members is either a BatchedCMAES (blocks (S, P, d) out, fitness (S, P) in) or a
    # blocked instance of the DE family / SepCMAES (candidates (P, S·d) out, fitness (P, S) in).
    def _batched_ask(self, state, first):
        """``blocks (S, P, d), the same side by side (P, S·d), state``"""
        out, state = member_ask(self.batched, state, first)
        if hasattr(self.batched, "as_candidates"):
            return out, self.batched.as_candidates(out), state
        return out.reshape(out.shape[0], self.num_subpops, self.sub_dim).permute(1, 0, 2), out, state

    def _batched_tell(self, state, fit, first):
        if not hasattr(self.batched, "as_candidates") and fit.dim() == 2:
            fit = fit.T.contiguous()
        return member_tell(self.batched, state, fit, first)

    def init_ask(self, state):
        if self.batched is not None:
            _, wide, state = self._batched_ask(state, True)
            init_pop = self._scatter(state, wide)
            return init_pop, state.replace(coop_pops=init_pop)
        pops = []
        for m in self.base_algorithms:
            pop, state = member_ask(m, state, True)
            pops.append(pop)
        init_pop = self._scatter(state, torch.cat(pops, dim=1))
        return init_pop, state.replace(coop_pops=init_pop)

    def init_tell(self, state, fitness):
        for m in self.base_algorithms:
            state = member_tell(m, state, fitness, True)
        best = torch.argmin(fitness)
        best_dec = self._gather(state, state.coop_pops[best])
        return state.replace(best_fit=fitness[best].expand(self.num_subpops).clone(), best_dec=best_dec, coop_pops=None)


class VectorizedCoevolution(MemberRecording, _CoevolutionBase):
    members_attr = "base_algorithms"

    _can_batch = True

    def setup(self, key):
        return self._base_state(key)

    def ask(self, state):
        if self.batched is not None:
            blocks, _, state = self._batched_ask(state, False)               # (S, P, d)
            pop_size = blocks.shape[1]
            coop = state.best_dec.expand(self.num_subpops, pop_size, self.dim).clone()
            # block i of the i-th group of context vectors: the diagonal of a (S, P, S, d) view
            diag = coop.view(self.num_subpops, pop_size, self.num_subpops, self.sub_dim).diagonal(dim1=0, dim2=2)   # (P, d, S)
            diag.copy_(blocks.permute(1, 2, 0))
            coop = self._scatter(state, coop.reshape(self.num_subpops * pop_size, self.dim))
            return coop, state.replace(coop_pops=coop)
        subs = []
        for m in self.base_algorithms:
            pop, state = member_ask(m, state, False)
            subs.append(pop)
        pop_size = subs[0].shape[0]
        coop = state.best_dec.expand(self.num_subpops, pop_size, self.dim).clone()
        for i, pop in enumerate(subs):
            coop[i, :, i * self.sub_dim:(i + 1) * self.sub_dim] = pop
        coop = self._scatter(state, coop.reshape(self.num_subpops * pop_size, self.dim))
        return coop, state.replace(coop_pops=coop)

    def tell(self, state, fitness):
        fit = fitness.reshape(self.num_subpops, -1)
        pop_size = fit.shape[1]
        if self.batched is not None:
            state = self._batched_tell(state, fit, False)
        else:
            for i, m in enumerate(self.base_algorithms):
                state = member_tell(m, state, fit[i], False)
        min_fit, arg = fit.min(dim=1)
        coop = state.coop_pops.reshape(self.num_subpops, pop_size, self.dim)
        best_rows = self._gather(state, coop[torch.arange(self.num_subpops, device=coop.device), arg])   # (S, dim)
        blocks = best_rows.reshape(self.num_subpops, self.num_subpops, self.sub_dim)
        own_block = torch.diagonal(blocks, dim1=0, dim2=1).T                                           # (S, sub_dim)
        improved = state.best_fit > min_fit
        best_dec = torch.where(improved[:, None], own_block, state.best_dec.reshape(self.num_subpops, self.sub_dim))
        return state.replace(best_dec=best_dec.reshape(self.dim), best_fit=torch.minimum(state.best_fit, min_fit),
                             coop_pops=None)


class Coevolution(_CoevolutionBase):
    def setup(self, key):
        return self._base_state(key, iter_counter=0)

    def ask(self, state):
        i = state.iter_counter % self.num_subpops
        pop, state = member_ask(self.base_algorithms[i], state, False)
        coop = state.best_dec.expand(pop.shape[0], self.dim).clone()
        coop[:, i * self.sub_dim:(i + 1) * self.sub_dim] = pop
        coop = self._scatter(state, coop)
        return coop, state.replace(coop_pops=coop)

    def tell(self, state, fitness):
        i = state.iter_counter % self.num_subpops
        state = member_tell(self.base_algorithms[i], state, fitness, False)
        min_fit, arg = fitness.min(dim=0)
        candidate = self._gather(state, state.coop_pops[arg])
        improved = state.best_fit[i] > min_fit
        best_dec = torch.where(improved, candidate, state.best_dec)
        best_fit = state.best_fit.clone()
        best_fit[i] = torch.minimum(best_fit[i], min_fit)
        return state.replace(best_dec=best_dec, best_fit=best_fit, iter_counter=state.iter_counter + 1, coop_pops=None)
