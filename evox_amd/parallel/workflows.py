"""Multi-GPU workflows: one process per GPU, collectives over RCCL (``gloo`` on CPU).

``ShardedWorkflow``   one algorithm instance spread over all ranks.  If the algorithm implements
                      the row-sharded protocol (``ask_sharded`` / ``tell_sharded``) each rank
                      generates, evaluates and reduces only its rows; otherwise the replicated
                      mode of ``StdWorkflow.enable_multi_devices`` is used (full ``ask`` on every
                      rank, sliced evaluation, fitness all-gather, full ``tell``).
``IslandWorkflow``    one *independent* population per rank; every ``migrate_every`` generations
                      each island's best individuals are all-gathered and offered to
                      ``algorithm.migrate`` (the hook the reference leaves unimplemented,
                      ``std_workflow.py:230-244``).
"""
from __future__ import annotations

from typing import Optional, Sequence

import os

import torch

from ..core import distributed as dist
from ..core import random as rnd
from ..core import tree as _tree
from ..core.algorithm import has_init_ask, has_init_tell
from ..core.module import use_state
from ..core.state import State
from ..workflows.std_workflow import StdWorkflow


def supports_sharding(algorithm) -> bool:
    return callable(getattr(algorithm, "ask_sharded", None)) and callable(getattr(algorithm, "tell_sharded", None))


class ShardedWorkflow(StdWorkflow):
    """``resync_every = k > 0``: every k generations rank 0's copy of the replicated algorithm state (every floating
    tensor of the algorithm's state except the row-local ones, ``algorithm.sharded_fields``, default
    ``("population",)``) is broadcast to the other ranks.  The replicas are computed from identical all-reduced
    statistics, but by library GEMMs whose algorithm each rank picks by its own timing, so they may differ in the
    last bits; the differences stay at rounding level (the covariance update contracts them), and the broadcast,
    a few MB every k generations, removes them altogether.

    ``monitor_candidates`` (row-sharded mode; in replicated mode every rank holds all candidates anyway):
    ``"full"`` – the candidate rows are all-gathered each generation (pop × dim elements) and the ``post_eval`` /
    ``pre_tell`` hooks get the full set, row-aligned with the fitness, so an
    :class:`~evox_amd.monitors.EvalMonitor` on rank 0 tracks the best individual of the whole population;
    ``"local"`` – the hooks get this rank's rows next to the full fitness (a monitor then sees the best among its own
    rows only); ``"auto"`` (default) – ``"full"`` when ANY rank has a monitor on those hooks, ``"local"`` otherwise:
    the ranks agree on it with one small all-reduce in ``init`` (the gather is a collective, so every rank has to take
    part whether or not it has monitors itself).  An explicit value must be the same on every rank.

    State fields declared with ``pytree_field(sharding=ShardingType.SHARED_FIRST_DIM)`` in the algorithm's state
    dataclass (or listed in ``algorithm.sharded_fields``) are row-local: ``resync`` leaves them alone and
    ``shard_state`` cuts them to this rank's rows."""

    def __init__(self, algorithm, problem, monitors=(), opt_direction="min", candidate_transforms=(),
                 fitness_transforms=(), resync_every: int = 0, monitor_candidates: str = "auto",
                 global_fitness_transforms=(), **kwargs):
        super().__init__(algorithm, problem, monitors, opt_direction, candidate_transforms, fitness_transforms, **kwargs)
        self.row_sharded = supports_sharding(algorithm)
        self.pmap_axis_name = dist.POP_AXIS_NAME
        self.resync_every = int(resync_every)
        if monitor_candidates not in ("local", "full", "auto"):
            raise ValueError("monitor_candidates must be 'auto', 'local' or 'full'")
        self.monitor_candidates = monitor_candidates
        # applied to the gathered (whole-population) fitness on every rank, before ``fitness_transforms``
        self.global_fitness_transforms = list(global_fitness_transforms)
        if self.global_fitness_transforms:
            self.fitness_transforms = self.global_fitness_transforms + self.fitness_transforms
        d = self.opt_direction
        self._minimising = bool((d == 1).all()) if isinstance(d, torch.Tensor) else d == 1
        self._presorted = None

    # -- distributed ranking ---------------------------------------------------------------
    # An algorithm whose ``tell_sharded`` carries ``accepts_presorted = True`` ranks the population without any process
    # sorting more than its own rows: every process sorts its fitness values BEFORE the all-gather and sends them along
    # (one message of 2 x local floats instead of local), and the global rank of a local row is its local rank plus one
    # binary search per other process.  A full sort of the gathered vector on every rank — replicated work that grows
    # with the number of processes — is what this replaces.  It needs the order of the raw fitness to be the order the
    # algorithm sees: minimisation and no fitness transforms.
    def _presort(self, fitness) -> bool:
        if os.environ.get("EVOX_AMD_SHARDED_PRESORT", "1") == "0":
            return False
        fn = getattr(self.algorithm, "tell_sharded", None)
        return (self.row_sharded and getattr(fn, "accepts_presorted", False) and self._minimising
                and not self.fitness_transforms and isinstance(fitness, torch.Tensor) and fitness.is_cuda
                and fitness.dim() == 1 and fitness.dtype == torch.float32)

    def _gather_fitness(self, fitness, n_total=None, world=1):
        self._presorted = None
        if not self._presort(fitness):
            return super()._gather_fitness(fitness, n_total, world)
        from ..ops import sort as _sort

        local = fitness.shape[0]
        msg = torch.empty((2, local), dtype=torch.float32, device=fitness.device)
        _, own_rank = _sort.sort_with_rank(fitness, out=msg[1], copy_out=msg[0])
        full = dist.all_gather(msg, self.pmap_axis_name, axis=0, tiled=True) if world > 1 else msg
        self._presorted = (full, own_rank)
        return full.view(world, 2, local)[:, 0].reshape(-1)

    def _local_fields(self, alg_state) -> set:
        """Names of the row-local fields of the algorithm state: ``algorithm.sharded_fields`` plus every field whose
        declaration carries ``sharding=ShardingType.SHARED_FIRST_DIM``."""
        import dataclasses

        local = set(getattr(self.algorithm, "sharded_fields", ("population",)))
        fields = getattr(alg_state, "_state_dict", None)
        if fields is not None and dataclasses.is_dataclass(fields):
            for f in dataclasses.fields(fields):
                if f.metadata.get("sharding") is dist.ShardingType.SHARED_FIRST_DIM:
                    local.add(f.name)
        return local

    def shard_state(self, state):
        """Cut every row-local field of the algorithm state that still has its full first dimension to this rank's
        rows (a replicated state, e.g. a single-process checkpoint, becomes this rank's share)."""
        alg = state.get_child_state("algorithm")
        rank, world = state.rank, state.world_size
        if world == 1:
            return state
        fresh = {}
        for name in self._local_fields(alg):
            if name in alg:
                v = getattr(alg, name)
                if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] % world == 0:
                    fresh[name] = dist.ShardingType.SHARED_FIRST_DIM.shard(v, rank, world)
        return state.replace_child("algorithm", alg.replace(**fresh)) if fresh else state

    def resync(self, state):
        """Rank 0's replicated algorithm state on every rank (collective: all ranks call it)."""
        alg = state.get_child_state("algorithm")
        local = self._local_fields(alg)
        fresh = {}
        for name in alg.keys():
            v = getattr(alg, name)
            if isinstance(v, torch.Tensor) and v.is_floating_point() and name not in local:
                fresh[name] = dist.broadcast(v, src=0)
        return state.replace_child("algorithm", alg.replace(**fresh)) if fresh else state

    def init(self, key=None, no_state: bool = False):
        state = super().init(key, no_state)
        self._resolve_monitor_candidates()
        if state is None:
            return None
        return state.replace(rank=dist.get_process_id(), world_size=dist.get_world_size())

    def _resolve_monitor_candidates(self) -> None:
        if self.monitor_candidates != "auto":
            return
        mine = bool(self.registered_hooks["post_eval"] or self.registered_hooks["pre_tell"])
        if dist.get_world_size() > 1:
            flag = torch.tensor([1.0 if mine else 0.0])
            if dist._dist().get_backend() != "gloo":
                flag = flag.cuda()
            mine = bool(float(dist.all_reduce(flag, "max")[0]) > 0)
        self.monitor_candidates = "full" if mine else "local"

    # -- checkpoints -----------------------------------------------------------------------
    def _local_keys(self, state):
        own = state.find_path_to(self._node_id, None)
        prefix = "/".join(f"children/{p}" for p in own[0]) if own and own[0] else ""
        alg = state.get_child_state("algorithm")
        base = (prefix + "/" if prefix else "") + "children/algorithm/fields/"
        return {base + name for name in self._local_fields(alg)
                if name in alg and isinstance(getattr(alg, name), torch.Tensor)}

    def save_checkpoint(self, state, path) -> None:
        """Collective.  The replicated state once (rank 0), each rank's row-local fields in a file of its own
        (``core/checkpoint.py``); a single-process job writes the same layout with one rank file."""
        from ..core.checkpoint import save_state_sharded

        local = self._local_keys(state) if (self.row_sharded and state.world_size > 1) else set()
        save_state_sharded(state, path, local, state.rank, state.world_size, dist.barrier)

    def load_checkpoint(self, path, template=None, device=None):
        """This rank's state from ``save_checkpoint`` — also when the job now has a different number of ranks (the
        row-local fields are re-cut; rank and world size are those of the loading job)."""
        from ..core.checkpoint import load_state_sharded

        rank, world = dist.get_process_id(), dist.get_world_size()
        state = load_state_sharded(path, rank, world, template, device)
        return state.replace(rank=rank, world_size=world)

    def _step(self, state, hooks=None, aux=None):
        # a population that the ranks cannot share evenly runs in replicated mode (remainder-aware evaluation slices)
        pop_size = getattr(self.algorithm, "pop_size", None)
        uneven = isinstance(pop_size, int) and state.world_size > 1 and pop_size % state.world_size != 0
        if not self.row_sharded or uneven:
            return super()._step(state, hooks, aux)
        hooks = self.registered_hooks if hooks is None else hooks
        rank, world = state.rank, state.world_size
        if self.monitor_candidates == "auto":              # a state that did not come from this object's ``init``
            self._resolve_monitor_candidates()
        for m in hooks["pre_step"]:
            m.pre_step(state)
        for m in hooks["pre_ask"]:
            m.pre_ask(state)
        cands, state = use_state(self.algorithm.ask_sharded)(state, rank, world)
        for m in hooks["post_ask"]:
            m.post_ask(state, cands)
        tcands = cands
        for t in self.candidate_transforms:
            tcands = t(tcands)
        for m in hooks["pre_eval"]:
            m.pre_eval(state, cands, tcands)
        fitness, state = self._evaluate(state, tcands)          # includes the fitness all-gather
        seen = cands
        if self.monitor_candidates == "full" and world > 1:
            seen = _tree.tree_map(lambda t: dist.all_gather(t, dist.POP_AXIS_NAME, axis=0, tiled=True), cands)
        for m in hooks["post_eval"]:
            m.post_eval(state, seen, tcands, fitness)
        tfitness = fitness
        for t in self.fitness_transforms:
            tfitness = t(tfitness)
        for m in hooks["pre_tell"]:
            m.pre_tell(state, seen, tcands, fitness, tfitness)
        extra = {} if self._presorted is None else {"presorted": self._presorted}
        self._presorted = None
        state = use_state(self.algorithm.tell_sharded)(state, tfitness, rank, world, **extra)
        if aux is not None:
            aux.update(cands=cands, tcands=tcands, fitness=fitness, tfitness=tfitness)
        for m in hooks["post_tell"]:
            m.post_tell(state)
        state = state.replace(generation=state.generation + 1, first_step=False)
        if self.resync_every > 0 and state.world_size > 1 and state.generation % self.resync_every == 0:
            state = self.resync(state)
        for m in hooks["post_step"]:
            m.post_step(state)
        return state


class IslandWorkflow(StdWorkflow):
    """Island model.  Every rank derives its own key (``fold_in(key, rank)``), so islands
    explore independently.  Migration: all-gather of each island's ``num_migrants`` best
    ``(individual, fitness)`` pairs — ``W · k · (dim + 1)`` floats, a latency-bound message —
    after which every island offers the *other* islands' migrants to ``algorithm.migrate``.

    The algorithm must expose ``population`` / ``fitness`` in its state and a
    ``migrate(state, pop, fit) -> state`` method."""

    def __init__(self, algorithm, problem, monitors=(), opt_direction="min", migrate_every: int = 10,
                 num_migrants: int = 8, topology: str = "all", **kwargs):
        super().__init__(algorithm, problem, monitors, opt_direction, **kwargs)
        if not callable(getattr(algorithm, "migrate", None)):
            raise TypeError(f"{type(algorithm).__name__} has no migrate(state, pop, fit) method")
        if topology not in ("all", "ring"):
            raise ValueError("topology must be 'all' or 'ring'")
        self.migrate_every, self.num_migrants, self.topology = migrate_every, num_migrants, topology

    def init(self, key=None, no_state: bool = False):
        rank = dist.get_process_id()
        state = super().init(rnd.fold_in(key, rank) if key is not None else None, no_state)
        if state is None:
            return None
        return state.replace(rank=rank)

    # names of the state fields that hold the individuals an island offers and how good they are
    _MIGRANT_FIELDS = (("population", "fitness"), ("local_best_location", "local_best_fitness"))

    def _migrate(self, state):
        """One migration: every island's ``num_migrants`` best individuals in one all-gather of ``k x (dim + 1)`` floats
        (the peer-to-peer one-shot form when it is switched on: one put and one collect launch), the immigrants of an
        island chosen from the other islands' packets by the hand-written radix select (``ops.sort.select_indices``: one
        launch, no sort)."""
        from ..ops import sort as _sort

        alg = state.get_child_state("algorithm")
        k = self.num_migrants
        for pop_name, fit_name in self._MIGRANT_FIELDS:
            if pop_name in alg and fit_name in alg:
                pop, fit = getattr(alg, pop_name), getattr(alg, fit_name)
                break
        else:
            raise TypeError(f"{type(self.algorithm).__name__}'s state has no population / fitness pair to migrate from")
        best = _sort.select_indices(fit, k)
        packet = torch.cat([pop[best], fit[best, None]], dim=1)                          # (k, dim + 1)
        world = dist.get_world_size()
        everyone = dist.all_gather(packet, dist.POP_AXIS_NAME, axis=0, tiled=True).reshape(world, k, -1)
        rank = dist.get_process_id()
        if self.topology == "ring":
            src = everyone[(rank - 1) % world]
        else:
            keep = [r for r in range(world) if r != rank] or [rank]
            flat = everyone[keep].reshape(-1, everyone.shape[-1])
            src = flat if flat.shape[0] <= k else flat[_sort.select_indices(flat[:, -1].contiguous(), k)]
        return use_state(self.algorithm.migrate)(state, src[:, :-1].contiguous(), src[:, -1].contiguous())

    def step(self, state):
        # migration stays outside the (optionally graph-replayed) generation: it is a collective
        # driven by the host-side generation counter
        state = super().step(state)
        if self.migrate_every > 0 and state.generation % self.migrate_every == 0:
            state = self._migrate(state)
        return state
