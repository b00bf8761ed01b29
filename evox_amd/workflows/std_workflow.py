"""The standard ask → evaluate → tell driver.

One ``step`` = monitor hooks, ``algorithm.ask`` (``init_ask`` on the first step), optional
population slicing when several ranks share the evaluation, candidate transforms,
``problem.evaluate``, fitness all-gather, fitness transforms, ``algorithm.tell`` and an
optional migration.  There is no tracing compiler: the step is ordinary Python that launches
fused kernels; ``graph_step=True`` records the steady-state step into a HIP graph and replays
it (``evox_amd.utils.graph``).

Multi-GPU: one process per GPU.  After ``enable_multi_devices`` every rank runs the same
algorithm on replicated state (identical keys ⇒ identical populations), evaluates its slice of the
rows and all-gathers the fitness over RCCL — the data-parallel mode of the reference
(``src/evox/workflows/std_workflow.py:33-313``).  The slices are remainder-aware (the first
``n mod W`` ranks take one row more, as the reference's actor workers do, ``workflows/distributed.py:49-54``;
its pmap mode refuses such sizes).  The ``post_eval`` / ``pre_tell`` hooks receive the full candidate
set, row-aligned with the full fitness they come with, and this rank's transformed slice.
"""
from __future__ import annotations

from collections.abc import Callable, Sequence
from typing import Optional, Union

import torch

from ..core import tree as _tree
from ..core.algorithm import Algorithm, has_init_ask, has_init_tell
from ..core.distributed import (POP_AXIS_NAME, all_gather, get_process_id, get_world_size, is_dist_initialized)
from ..core.module import use_state
from ..core.monitor import HOOK_NAMES, Monitor
from ..utils import profiling as _prof
from ..core.problem import Problem
from ..core.state import State
from ..core.struct import dataclass, pytree_field
from ..core.workflow import Workflow
from ..utils.common import parse_opt_direction


@dataclass
class StdWorkflowState:
    generation: int
    first_step: bool = pytree_field(static=True)
    rank: int = pytree_field(static=True)
    world_size: int = pytree_field(static=True)


_NO_HOOKS = {name: () for name in HOOK_NAMES}


class StdWorkflow(Workflow):
    def __init__(
        self,
        algorithm: Algorithm,
        problem: Problem,
        monitors: Sequence[Monitor] = (),
        opt_direction: Union[str, Sequence[str]] = "min",
        candidate_transforms: Sequence[Callable] = (),
        fitness_transforms: Sequence[Callable] = (),
        jit_step: bool = True,
        external_problem: bool = False,
        num_objectives: Optional[int] = None,
        migrate_helper=None,
        graph_step: bool = False,
        graph_mode: str = "copy",
    ):
        """
        algorithm, problem      the two halves of the loop.
        monitors                called in list order at the hooks they declare.
        opt_direction           "min" / "max", or one per objective; fitness is multiplied by
                                ±1 so that algorithms always minimise.
        candidate_transforms    applied in order to the candidates before evaluation
                                (e.g. vector → parameter tree).
        fitness_transforms      applied in order to the fitness before ``tell``.
        jit_step                accepted for source compatibility; ignored.
        external_problem        the problem runs on the host (simulator, numpy): candidates are
                                copied to host memory, the fitness is copied back.  Requires
                                ``num_objectives``.
        migrate_helper          object with ``migrate_from_human() -> (do, pop, fit)``; when
                                ``do`` is true ``algorithm.migrate(state, pop, fit)`` is called.
        graph_step              replay the steady-state step from a HIP graph (GPU only).  ``"auto"``: decide by
                                measurement — after the algorithm's settling generations, 16 generations are issued
                                eagerly and 16 replayed, and the faster way is kept (a step that cannot be recorded
                                stays eager); the timings end up in ``graph_choice``.  Replays win where a generation
                                is many small kernels; a generation that already keeps the GPU busy (full-matrix
                                CMA-ES) is faster issued eagerly.
        graph_mode              "copy" (default): value semantics like the reference's compiled step — every replay
                                copies the state in and clones it out, so a state the caller keeps (a history list, a
                                best-so-far snapshot, a deferred checkpoint) stays what it was.  "ring": pointer-stable
                                replay for benchmarks and tight loops — the state tensors handed out are the recordings'
                                own buffers, valid for one further step (clone what must live longer); monitors with
                                ``post_tell`` / ``post_step`` hooks are handed a snapshot (``utils.graph``).
        """
        super().__init__()
        self.algorithm = algorithm
        self.problem = problem
        self.monitors = list(monitors)
        self.registered_hooks = {name: [] for name in HOOK_NAMES}
        for m in self.monitors:
            for h in m.hooks():
                if h not in self.registered_hooks:
                    raise ValueError(f"unknown monitor hook {h!r}")
                self.registered_hooks[h].append(m)
        self.opt_direction = parse_opt_direction(opt_direction)
        for m in self.monitors:
            m.set_opt_direction(self.opt_direction)
        self.candidate_transforms = list(candidate_transforms)
        self.fitness_transforms = list(fitness_transforms)
        self.jit_step = jit_step
        self.external_problem = external_problem
        self.num_objectives = num_objectives
        self.migrate_helper = migrate_helper
        self.graph_step = graph_step
        self._variants_issued = set()
        self.graph_mode = graph_mode
        self.graph_replays = 0
        self._eager_before_graph = 0
        self._auto = None
        self.graph_choice = None
        if external_problem and num_objectives is None:
            raise ValueError("Using external problem, but num_objectives isn't set")
        self.pmap_axis_name = None
        self._graphed = None

    # ------------------------------------------------------------------------------------
    def setup(self, key):
        return State(StdWorkflowState(generation=0, first_step=True, rank=0, world_size=1))

    def _ask(self, state):
        first = has_init_ask(self.algorithm) and state.first_step
        ask = self.algorithm.init_ask if first else self.algorithm.ask
        return use_state(ask)(state)

    @staticmethod
    def slice_bounds(n: int, rank: int, world: int):
        """Rows ``[lo, hi)`` of rank ``rank`` when ``n`` rows are spread over ``world`` ranks."""
        per, rem = divmod(n, world)
        lo = per * rank + min(rank, rem)
        return lo, lo + per + (1 if rank < rem else 0)

    def _gather_fitness(self, fitness, n_total=None, world=1):
        if n_total is None or world == 1 or n_total % world == 0:
            return all_gather(fitness, self.pmap_axis_name, axis=0, tiled=True)
        # uneven slices: pad to the largest, gather, drop the padding rows
        per = n_total // world + 1
        pad = per - fitness.shape[0]
        if pad:
            fitness = torch.cat([fitness, fitness.new_zeros((pad,) + tuple(fitness.shape[1:]))], dim=0)
        full = all_gather(fitness, self.pmap_axis_name, axis=0, tiled=True)
        parts = []
        for r in range(world):
            lo, hi = self.slice_bounds(n_total, r, world)
            parts.append(full[r * per:r * per + (hi - lo)])
        return torch.cat(parts, dim=0)

    def _evaluate(self, state, cands, n_total=None):
        if not self.external_problem:
            fitness, state = use_state(self.problem.evaluate)(state, cands)
        else:
            leaves = _tree.tree_leaves(cands)
            dev = leaves[0].device
            host_cands = _tree.tree_map(lambda t: t.detach().cpu() if isinstance(t, torch.Tensor) else t, cands)
            fitness, state = use_state(self.problem.evaluate)(state, host_cands)
            fitness = torch.as_tensor(fitness, dtype=torch.float32)
            n = leaves[0].shape[0]
            want = (n,) if self.num_objectives == 1 else (n, self.num_objectives)
            if tuple(fitness.shape) != want:
                raise ValueError(f"external problem returned fitness of shape {tuple(fitness.shape)}, expected {want}")
            fitness = fitness.to(dev)
        fitness = self._gather_fitness(fitness, n_total, getattr(state, "world_size", 1))
        d = self.opt_direction
        if isinstance(d, torch.Tensor):
            fitness = fitness * d.to(fitness.device)
        elif d != 1:
            fitness = fitness * d
        return fitness, state

    def _tell(self, state, fitness):
        first = has_init_tell(self.algorithm) and state.first_step
        tell = self.algorithm.init_tell if first else self.algorithm.tell
        return use_state(tell)(state, fitness)

    def _step(self, state, hooks=None, aux=None):
        """One generation.  ``hooks`` overrides the registered monitor hooks (graph capture
        records the step without them); ``aux`` (a dict) receives the intermediate tensors."""
        hooks = self.registered_hooks if hooks is None else hooks
        for m in hooks["pre_step"]:
            m.pre_step(state)
        for m in hooks["pre_ask"]:
            m.pre_ask(state)

        with _prof.scope("ask"):
            cands, state = self._ask(state)

        for m in hooks["post_ask"]:
            m.post_ask(state, cands)

        world = state.world_size
        all_cands, n_total = cands, None
        if world > 1:
            n_total = _tree.tree_leaves(cands)[0].shape[0]
            lo, hi = self.slice_bounds(n_total, state.rank, world)
            cands = _tree.tree_map(lambda x: x[lo:hi], cands)

        tcands = cands
        for t in self.candidate_transforms:
            tcands = t(tcands)

        for m in hooks["pre_eval"]:
            m.pre_eval(state, cands, tcands)

        with _prof.scope("evaluate"):
            fitness, state = self._evaluate(state, tcands, n_total)

        for m in hooks["post_eval"]:
            m.post_eval(state, all_cands, tcands, fitness)

        tfitness = fitness
        for t in self.fitness_transforms:
            tfitness = t(tfitness)

        for m in hooks["pre_tell"]:
            m.pre_tell(state, all_cands, tcands, fitness, tfitness)

        with _prof.scope("tell"):
            state = self._tell(state, tfitness)
        if aux is not None:
            aux.update(cands=all_cands, tcands=tcands, fitness=fitness, tfitness=tfitness)

        for m in hooks["post_tell"]:
            m.post_tell(state)

        if self.migrate_helper is not None:
            do_migrate, foreign_pop, foreign_fit = self.migrate_helper.migrate_from_human()
            if bool(do_migrate):
                state = use_state(self.algorithm.migrate)(state, foreign_pop, foreign_fit)

        if state.first_step:
            state = state.replace(generation=state.generation + 1, first_step=False)
        else:
            state = state.replace(generation=state.generation + 1)

        for m in hooks["post_step"]:
            m.post_step(state)
        return state

    def _bare_step(self, state):
        aux = {}
        state = self._step(state, hooks=_NO_HOOKS, aux=aux)
        return state, aux

    _AUTO_GENERATIONS = 16           # generations timed each way by graph_step="auto"

    def step(self, state):
        if not (self.graph_step and not state.first_step):
            return self._step(state)
        if self.graph_step == "auto":
            return self._step_auto(state)
        return self._step_graphed(state)

    def _step_auto(self, state):
        """``graph_step="auto"``: settle, time eager generations, record, time replays, keep the faster way.  Every call
        advances the run by one generation; the four synchronisations of the measurement are its only cost."""
        import time

        a = self._auto
        if a is None:
            on_gpu = any(isinstance(t, torch.Tensor) and t.is_cuda for t in _tree.tree_leaves(state))
            if not on_gpu or self.migrate_helper is not None or self.external_problem:
                self.graph_step = False
                self.graph_choice = {"mode": "eager", "reason": "nothing to record (host-side step or no GPU)"}
                return self._step(state)
            settle = max(4, int(getattr(self.algorithm, "graph_settle_steps", 0) or 0))
            a = self._auto = {"phase": "settle", "count": 0, "settle": settle}

        def clock():
            torch.cuda.synchronize()
            return time.perf_counter()

        n = self._AUTO_GENERATIONS
        if a["phase"] == "settle":
            a["count"] += 1
            out = self._step(state)
            if a["count"] >= a["settle"]:
                a.update(phase="eager", count=0, t0=clock())
            return out
        if a["phase"] == "eager":
            out = self._step(state)
            a["count"] += 1
            if a["count"] >= n:
                a.update(phase="record", count=0, eager_ms=(clock() - a["t0"]) / n * 1e3)
                self._eager_before_graph = a["settle"]           # the settling generations are behind us
            return out
        try:
            out = self._step_graphed(state)
        except Exception as e:                                   # the step cannot be recorded: eager it is
            from ..utils.graph import GraphCaptureError

            if not isinstance(e, (GraphCaptureError, ValueError)):
                raise
            self.graph_step, self._graphed = False, None
            self.graph_choice = {"mode": "eager", "eager_ms": a["eager_ms"], "reason": str(e).splitlines()[0]}
            return self._step(state)
        a["count"] += 1
        if a["phase"] == "record":                               # the recording and its first replays are not timed
            if a["count"] >= 4:
                a.update(phase="graph", count=0, t0=clock())
            return out
        if a["count"] >= n:
            graph_ms = (clock() - a["t0"]) / n * 1e3
            keep = graph_ms < 0.97 * a["eager_ms"]
            self.graph_choice = {"mode": "graph" if keep else "eager", "eager_ms": a["eager_ms"], "graph_ms": graph_ms}
            self.graph_step = bool(keep)
            if not keep:
                self._graphed = None
        return out

    def _step_graphed(self, state):
        # An algorithm whose host-side control settles over its first generations (the iteration count of CMA-ES's
        # factorisation starts conservative) asks for that many eager generations before the first recording: in eager
        # mode the control adapts every generation, in a recording it is frozen between polls.
        settle = int(getattr(self.algorithm, "graph_settle_steps", 0) or 0)
        if self._graphed is None and self._eager_before_graph < settle:
            self._eager_before_graph += 1
            return self._step(state)
        # Steady-state generations replay a recording of the bare step; the monitors observe
        # it from outside: pre_* hooks see the incoming state, post_* hooks the outgoing one,
        # with the candidates / fitness the recording produced.
        # An algorithm whose step branches on host counters (a phase switch, a periodic regrouping) names the
        # branch taken from the incoming state: one recording per variant.
        variant_of = getattr(self.algorithm, "graph_variant", None)
        variant = variant_of(state.get_child_state("algorithm")) if variant_of is not None else None
        if self._graphed is None:
            self._graphed = {}
        graphed = self._graphed.get(variant)
        if graphed is None and variant not in self._variants_issued:
            # The first generation of a variant is issued eagerly: whatever its kernels set up at first use with these
            # shapes (GEMM plans chosen by timing, workspaces, a kernel's LDS limit) involves the host and cannot be
            # recorded — e.g. the doubled population after a restart of IPOP-CMA-ES.
            self._variants_issued.add(variant)
            return self._step(state)
        if graphed is None:
            from ..utils.graph import GraphedStep

            if self.migrate_helper is not None or self.external_problem:
                raise ValueError("graph_step cannot record host-side evaluation or migration")
            extra = getattr(self.algorithm, "graph_host_counters", None)
            counters = ["fields/generation"]
            for c in (extra() if extra else ()):         # names, or (name, modulus) for counters that wrap around;
                name = c if isinstance(c, str) else c[0]  # a name with a "/" is a path below the algorithm's state (containers)
                path = f"children/algorithm/{name}" if "/" in name else f"children/algorithm/fields/{name}"
                counters.append(path if isinstance(c, str) else (path, c[1]))
            # an algorithm that asks for new recordings while it runs (the factorisation control of CMA-ES) keeps the
            # short ring: recording is the expensive part there, and it is paid per recording of the ring
            rerecords = any(callable(getattr(self.algorithm, h, None)) for h in ("graph_retune", "graph_check"))
            graphed = self._graphed[variant] = GraphedStep(self._bare_step, host_counters=counters, mode=self.graph_mode,
                                                           ring=2 if rerecords else None)
        hooks = self.registered_hooks
        for m in hooks["pre_step"]:
            m.pre_step(state)
        for m in hooks["pre_ask"]:
            m.pre_ask(state)
        new_state, aux = graphed(state)
        self.graph_replays += 1
        if any(hooks[h] for h in ("post_ask", "pre_eval", "post_eval", "pre_tell")):
            # the recording's own buffers are overwritten a few replays later: monitors get copies they may keep
            aux = {k: _tree.tree_map(lambda t: t.clone() if isinstance(t, torch.Tensor) else t, v) for k, v in aux.items()}
        # host decisions frozen into the recording are revisited now and then (often at first, while an iteration
        # control settles from its conservative start): the algorithm may hand back a state with new decisions, which
        # is then recorded afresh
        period = 8 if self.graph_replays <= 160 else 64
        if self.graph_replays % period == 0:
            retune = getattr(self.algorithm, "graph_retune", None)
            if retune is not None:
                tuned = retune(new_state.get_child_state("algorithm"))
                if tuned is not None:
                    new_state = new_state.replace_child("algorithm", tuned)
                    self._graphed = None
            else:
                check = getattr(self.algorithm, "graph_check", None)
                if check is not None and check():
                    self._graphed = None        # record again with the algorithm's updated settings
        for m in hooks["post_ask"]:
            m.post_ask(new_state, aux["cands"])
        for m in hooks["pre_eval"]:
            m.pre_eval(new_state, aux["cands"], aux["tcands"])
        for m in hooks["post_eval"]:
            m.post_eval(new_state, aux["cands"], aux["tcands"], aux["fitness"])
        for m in hooks["pre_tell"]:
            m.pre_tell(new_state, aux["cands"], aux["tcands"], aux["fitness"], aux["tfitness"])
        seen = new_state
        if self.graph_mode == "ring" and (hooks["post_tell"] or hooks["post_step"]):
            # ring replay hands out the recordings' own buffers, overwritten one step later: a monitor that keeps
            # what it is shown (PopMonitor's history) gets a snapshot instead
            seen = _tree.tree_map(lambda t: t.clone() if isinstance(t, torch.Tensor) else t, new_state)
        for m in hooks["post_tell"]:
            m.post_tell(seen)
        for m in hooks["post_step"]:
            m.post_step(seen)
        return new_state

    # ------------------------------------------------------------------------------------
    def enable_multi_devices(self, state: State, pmap_axis_name=POP_AXIS_NAME) -> State:
        """Switch to population-sharded evaluation over all ranks of the job.

        Call after ``init`` on every rank (same key everywhere).  Requires an initialised
        process group when the job has more than one process."""
        self.pmap_axis_name = pmap_axis_name
        own = state.find_path_to(self._node_id, None)
        world, rank = get_world_size(), get_process_id()
        if own is None:
            return state.replace(rank=rank, world_size=world)
        path, node = own
        return state.replace_by_path(path, node.replace(rank=rank, world_size=world))
