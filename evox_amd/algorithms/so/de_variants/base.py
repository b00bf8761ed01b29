"""Shared machinery of the differential-evolution family.

All variants keep the steady-state batch layout of the reference (``population, fitness,
best_index, start_index, key, trial_vectors``; a window of ``batch_size`` individuals is varied
per generation and slides over the population) and the greedy compare-and-replace ``tell``
(reference: ``src/evox/algorithms/so/de_variants/de.py:137-166``).  Variation is batched: one
call builds the trial vectors of the whole window.
"""
from __future__ import annotations

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.state import State
from ....ops import sort as _sort

# base-vector codes of the strategy tables (rand, best, pbest, current) and crossover codes
RAND, BEST, PBEST, CURRENT = 0, 1, 2, 3
BIN, EXP, ARITH = 0, 1, 2


class DEBase(Algorithm):
    strict_improvement = True       # replace when trial < parent (else <=)

    def _init_common(self, lb, ub, pop_size, batch_size=None):
        super().__init__()
        self.lb = torch.as_tensor(lb, dtype=torch.float32)
        self.ub = torch.as_tensor(ub, dtype=torch.float32).to(self.lb.device)
        if not bool((self.lb < self.ub).all()):
            raise ValueError("lb must be below ub in every dimension")
        self.dim = self.lb.shape[0]
        self.pop_size = int(pop_size)
        self.batch_size = int(batch_size) if batch_size else self.pop_size
        self.blocks = 1
        self.block_fitness = False

    def blocked(self, times: int, shared_fitness: bool = True):
        """``times`` independent copies of this algorithm, all told the same fitness, as ONE instance over the
        concatenated decision vector (what a decomposition container runs; the reference vmaps the member,
        ``src/evox/algorithms/containers/clustered_algorithm.py:10-42``).  A row of the population is ``times``
        sub-vectors; every sub-vector draws its own parents, crossover mask and control parameters, and keeps its own
        adaptation memory, while fitness, replacement decisions, the best index and the p-best set — functions of the
        shared fitness alone — exist once.  A generation costs the launches of one member.

        ``shared_fitness=False``: every block is told a fitness of its own (cooperative co-evolution,
        ``src/evox/algorithms/containers/coevolution.py:12-136``): ``tell`` takes ``(pop_size, times)`` values — or a
        vector, meaning the same value for every block —, ``fitness`` has that shape, ``best_index`` and the p-best set
        exist per block and every sub-vector is replaced on its own."""
        from ....core.module import clone_module

        if self.blocks != 1:
            raise ValueError("already a blocked instance")
        wide = clone_module(self)
        wide.lb, wide.ub = self.lb.repeat(times), self.ub.repeat(times)
        wide.dim, wide.blocks, wide.block_fitness = self.dim * times, int(times), not shared_fitness
        for name in ("mean", "stdvar"):
            v = getattr(self, name, None)
            if isinstance(v, torch.Tensor) and v.dim() == 1 and v.shape[0] == self.dim:
                setattr(wide, name, v.repeat(times))
        return wide

    def _init_population(self, key, mean=None, stdvar=None):
        dev = self.lb.device
        if mean is not None and stdvar is not None:
            pop = torch.clamp(stdvar * rnd.normal(key, (self.pop_size, self.dim)) + mean, self.lb, self.ub)
        else:
            pop = rnd.uniform(key, (self.pop_size, self.dim)) * (self.ub - self.lb) + self.lb
        return pop, torch.full((self.pop_size,), float("inf"), device=dev)

    def _base_state(self, key, trial_rows=None, **extra):
        dev = self.lb.device
        state_key, init_key = rnd.split(key.to(dev))
        pop, fit = self._init_population(init_key, getattr(self, "mean", None), getattr(self, "stdvar", None))
        if self.block_fitness:                       # one fitness column and one best index per block
            fit = fit[:, None].repeat(1, self.blocks)
        return State(population=pop, fitness=fit,
                     best_index=torch.zeros((self.blocks,) if self.block_fitness else (), dtype=torch.int64, device=dev),
                     start_index=0, start_index_dev=torch.zeros((), dtype=torch.int64, device=dev), key=state_key,
                     trial_vectors=torch.zeros((trial_rows or self.batch_size, self.dim), device=dev), **extra)

    def _window(self, state):
        """Rows varied this generation.  The offset is read from its device copy, so a recorded step follows
        the moving window when it is replayed."""
        if self.batch_size == self.pop_size:
            return torch.arange(self.pop_size, device=self.lb.device)
        return (torch.arange(self.batch_size, device=self.lb.device) + state.start_index_dev) % self.pop_size

    def _next_start_dev(self, state):
        return (state.start_index_dev + self.batch_size) % self.pop_size

    def graph_host_counters(self):
        return (("start_index", self.pop_size),)      # wraps around; the device work reads start_index_dev

    def _select(self, state, trial_vectors, trial_fitness):
        """Greedy replacement inside the window; returns (population, fitness, best_index,
        next start_index, success mask, parent fitness)."""
        idx = self._window(state)
        parent_fit = state.fitness[idx]
        if self.block_fitness:
            # every sub-vector competes with its own parent on its block's fitness: (rows, blocks) decisions
            B, b = self.blocks, idx.shape[0]
            if trial_fitness.dim() == 1:
                trial_fitness = trial_fitness[:, None].expand(b, B)
            better = trial_fitness < parent_fit if self.strict_improvement else trial_fitness <= parent_fit
            new_rows = torch.where(better[:, :, None], trial_vectors.reshape(b, B, -1),
                                   state.population[idx].reshape(b, B, -1)).reshape(b, self.dim)
            new_fit = torch.where(better, trial_fitness, parent_fit)
            if self.batch_size == self.pop_size:
                population, fitness = new_rows, new_fit
            else:
                population = state.population.index_copy(0, idx, new_rows)
                fitness = state.fitness.index_copy(0, idx, new_fit)
            start = (state.start_index + self.batch_size) % self.pop_size
            return population, fitness, torch.argmin(fitness, dim=0), start, better, parent_fit
        better = trial_fitness < parent_fit if self.strict_improvement else trial_fitness <= parent_fit
        new_rows = torch.where(better[:, None], trial_vectors, state.population[idx])
        new_fit = torch.where(better, trial_fitness, parent_fit)
        if self.batch_size == self.pop_size:
            population, fitness = new_rows, new_fit
        else:
            population = state.population.index_copy(0, idx, new_rows)
            fitness = state.fitness.index_copy(0, idx, new_fit)
        best_index = torch.argmin(fitness)
        start = (state.start_index + self.batch_size) % self.pop_size
        return population, fitness, best_index, start, better, parent_fit

    def migrate(self, state, foreign_population, foreign_fitness):
        """Island-model hook (every DE variant shares the population / fitness / best_index state): the immigrants
        replace the worst residents they improve on.  The reference only declares the hook
        (``src/evox/workflows/std_workflow.py:230-244``)."""
        from ....ops import sort as _sort

        if self.block_fitness:
            raise NotImplementedError("migration into an instance with one fitness per block is not defined")
        k = foreign_fitness.shape[0]
        worst = _sort.select_indices(state.fitness, k, largest=True)
        better = foreign_fitness < state.fitness[worst]
        pop = state.population.index_copy(0, worst, torch.where(better[:, None], foreign_population,
                                                                state.population[worst]))
        fit = state.fitness.index_copy(0, worst, torch.where(better, foreign_fitness, state.fitness[worst]))
        return state.replace(population=pop, fitness=fit, best_index=torch.argmin(fit))

    def tell(self, state, trial_fitness):
        population, fitness, best_index, start, _, _ = self._select(state, state.trial_vectors, trial_fitness)
        return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start,
                             start_index_dev=self._next_start_dev(state))

    # -- batched strategy evaluation -----------------------------------------------------
    def _strategy_trials(self, key, state, idx, prim, sec, num_diff, cross, F, CR, diff_padding_num, replace=False,
                         code_sets=None):
        """Trial vectors for window ``idx``.  ``prim``/``sec`` (base-vector codes), ``num_diff``
        and ``cross`` may be ints or per-row tensors; ``F``/``CR`` scalars or per-row tensors::

            base  = prim + F (sec − prim)
            v     = base + F Σ± differences
            trial = crossover(v, current; CR)   clipped to the bounds
        """
        from ....operators.crossover.differential_evolution import de_arith_recom, de_bin_cross, de_exp_cross

        pop, fit = state.population, state.fitness
        dev = pop.device
        from ....ops import de as _de
        static_codes = all(isinstance(v, int) for v in (prim, sec, num_diff, cross))
        n_picks_early = min(diff_padding_num, 2 * num_diff + 1) if isinstance(num_diff, int) else diff_padding_num
        fused = _de.supported(pop, prim, sec, num_diff, cross, n_picks_early, replace, code_sets)
        fused_keys = None
        if fused and static_codes:
            # every key of the fused launch in one split: (selection, p-best, crossover) and the crossover key's two children
            ks = rnd.split_nested(key, 3, 2, 2)
            k_sel, k_pb, k_x = ks[0], ks[1], ks[2]
            fused_keys = (ks[0], ks[3], ks[4])
        else:
            k_sel, k_pb, k_x = rnd.split(key, 3)
        # Blocked instance: everything below runs over virtual rows (individual, block) -> individual * B + block of
        # the population viewed as (pop_size * B, block_dim); parents are individuals, taken from the row's own block.
        B = self.blocks
        rows_out = idx.shape[0]
        if B > 1:
            blk = torch.arange(B, device=dev).repeat(rows_out)
            who = idx.repeat_interleave(B)                # the individual of every virtual row
            idx = who * B + blk
            pop_v = pop.reshape(self.pop_size * B, self.dim // B)

            def rows_of(ind):                             # rows of individuals `ind` (one per virtual row) in the row's block
                return pop_v[ind * B + (blk if ind.dim() == 1 else blk[:, None])]
        else:
            who, pop_v = idx, pop

            def rows_of(ind):
                return pop_v[ind]
        b = idx.shape[0]
        # Host-known settings stay on the host: no scalar ever crosses to the device here, so the
        # whole routine is free of synchronisations (and can be recorded into a graph).
        static_nd = isinstance(num_diff, int)
        n_picks = min(diff_padding_num, 2 * num_diff + 1) if static_nd else diff_padding_num

        def pbest_individuals():
            top = max(int(self.pop_size * 0.05), 1)
            pick = rnd.randint(k_pb, (b,), 0, top)
            if self.block_fitness:                             # the p-best set of every block, each in index order
                best = torch.topk(fit.T, top, dim=1, largest=False).indices.sort(dim=1).values       # (blocks, top)
                return best[blk, pick]
            best = _sort.select_indices(fit, top)              # the p-best set, in index order
            return best[pick]

        def finish(trial_v):
            return torch.clamp(trial_v.reshape(rows_out, self.dim), self.lb, self.ub)

        if fused:
            # GPU: one fused launch (same random elements as the formulation below)
            pbest_rows = pbest_individuals() if (not static_nd or PBEST in (prim, sec)) else None
            return _de.de_trial(key, pop, idx, state.best_index, pbest_rows, prim, sec, num_diff, F, CR, n_picks,
                                self.lb, self.ub, cross, blocks=B, keys=fused_keys).reshape(rows_out, self.dim)
        if replace:
            picks = rnd.randint(k_sel, (b, n_picks), 0, self.pop_size)
            picks = torch.where(picks == who[:, None], torch.full_like(picks, self.pop_size - 1), picks)
        else:
            picks = rnd.choice_distinct(k_sel, self.pop_size, b, n_picks, exclude=who)
        if static_nd:
            diff = rows_of(picks[:, 1]) - rows_of(picks[:, 2])
            for t in range(1, num_diff):
                diff = diff + (rows_of(picks[:, 2 * t + 1]) - rows_of(picks[:, 2 * t + 2]))
        else:
            nd = num_diff.to(dev).reshape(-1, 1)
            pos = torch.arange(n_picks, device=dev)[None, :]
            live = (pos >= 1) & (pos < 2 * nd + 1)
            sign = (torch.where(pos % 2 == 1, 1.0, -1.0) * live).to(pop.dtype)
            diff = torch.einsum("bk,bkd->bd", sign.expand(b, -1), rows_of(picks))
        current = pop_v[idx]

        def best_rows():
            if self.block_fitness:
                return rows_of(state.best_index[blk])
            return rows_of(state.best_index.expand(b)) if B > 1 else pop[state.best_index].expand(b, -1)

        def base_of(code):
            if isinstance(code, int):
                c = code
                if c == RAND:
                    return rows_of(picks[:, 0])
                if c == BEST:
                    return best_rows()
                if c == PBEST:
                    return rows_of(pbest_individuals())
                return current
            code_t = code.to(dev)
            cands = torch.stack([rows_of(picks[:, 0]), best_rows(), rows_of(pbest_individuals()), current], dim=1)
            return cands[torch.arange(b, device=dev), code_t]

        Fc = F[:, None] if isinstance(F, torch.Tensor) and F.dim() == 1 else F
        p, s = base_of(prim), base_of(sec)
        mutant = p + Fc * (s - p) + Fc * diff
        if isinstance(cross, int):
            c = cross
            if c == BIN:
                trial = de_bin_cross(k_x, mutant, current, CR)
            elif c == EXP:
                trial = de_exp_cross(k_x, mutant, current, CR)
            else:
                trial = de_arith_recom(mutant, current, CR)
        else:
            cross_t = cross.to(dev)
            kb, ke = rnd.split(k_x)
            t_bin = de_bin_cross(kb, mutant, current, CR)
            t_exp = de_exp_cross(ke, mutant, current, CR)
            t_ari = de_arith_recom(mutant, current, CR)
            trial = torch.where((cross_t == BIN)[:, None], t_bin, torch.where((cross_t == EXP)[:, None], t_exp, t_ari))
        return finish(trial)
