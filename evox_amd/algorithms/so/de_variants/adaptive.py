"""Self-adaptive DE variants: JaDE, SHADE, SaDE, CoDE.

Reference: ``de_variants/{jade,shade,sade,code}.py``.  The success-history statistics that the
reference accumulates with O(N) sequential loops only ever enter order-independent sums, so
they are single masked reductions here.
"""
from __future__ import annotations

import torch

from ....core import random as rnd
from ....ops import de_adapt as _adapt
from ....ops import sort as _sort
from .base import ARITH, BEST, BIN, CURRENT, EXP, PBEST, RAND, DEBase


def _nan_to(x, v=0.0):
    return torch.where(torch.isnan(x), torch.full_like(x, v), x)


class JaDE(DEBase):
    """JADE (Zhang & Sanderson 2009) without archive: current-to-pbest/1/bin,
    F ~ N(μF, 0.1), CR ~ Cauchy(μCR, 0.1), Lehmer / arithmetic mean adaptation."""

    strict_improvement = False

    def __init__(self, lb, ub, pop_size=100, diff_padding_num=9, differential_weight=None, cross_probability=None,
                 c=0.1):
        self._init_common(lb, ub, pop_size)
        self.diff_padding_num = diff_padding_num
        self.c = c
        self.num_diff_vects = 1

    # A blocked instance (``DEBase.blocked``) keeps one pair of means per block — ``F_u`` / ``CR_u`` of shape (blocks,) —
    # and control parameters per virtual row individual * blocks + block; its bookkeeping is the tensor formulation
    # with a block axis (the fused adaptation kernels handle the single-member layout).
    def setup(self, key):
        dev = self.lb.device
        shape = () if self.blocks == 1 else (self.blocks,)
        return self._base_state(key, F_u=torch.full(shape, 0.5, device=dev), CR_u=torch.full(shape, 0.5, device=dev),
                                F_vect=torch.zeros(self.pop_size * self.blocks, device=dev),
                                CR_vect=torch.zeros(self.pop_size * self.blocks, device=dev))

    def _draw_params(self, state, kF, kCR, loc_F, loc_CR):
        n = self.pop_size * self.blocks
        F = torch.clamp(rnd.normal(kF, (n,)) * 0.1 + loc_F, 0.0, 1.0)
        CR = torch.clamp(rnd.cauchy(kCR, (n,)) * 0.1 + loc_CR, 0.0, 1.0)
        return F, CR

    def _fused_tell_ok(self, state):
        # the fused kernels replace the whole population at once
        return (_adapt.supported(state.population) and self.batch_size == self.pop_size and self.blocks == 1)

    def ask(self, state):
        key, k_var, kF, kCR = rnd.split(state.key, 4)
        if self.blocks > 1:
            F, CR = self._draw_params(state, kF, kCR, state.F_u.repeat(self.pop_size), state.CR_u.repeat(self.pop_size))
        elif _adapt.supported(state.population):
            F, CR = _adapt.draw_params(kF, kF, kCR, state.F_u, state.CR_u, self.pop_size)
        else:
            F, CR = self._draw_params(state, kF, kCR, state.F_u, state.CR_u)
        idx = self._window(state)
        trial = self._strategy_trials(k_var, state, idx, CURRENT, PBEST, self.num_diff_vects, BIN, F, CR,
                                      self.diff_padding_num)
        return trial, state.replace(trial_vectors=trial, key=key, F_vect=F, CR_vect=CR)

    def tell(self, state, trial_fitness):
        if self._fused_tell_ok(state):
            population, fitness, best_index, F_u, CR_u, _ = _adapt.select_and_adapt(
                state.population, state.trial_vectors, state.fitness, trial_fitness, state.F_vect, state.CR_vect,
                state.F_u, state.CR_u, strict=self.strict_improvement, weight_by_gain=False, mode="jade", c=self.c)
            return state.replace(population=population, fitness=fitness, best_index=best_index, F_u=F_u, CR_u=CR_u)
        population, fitness, best_index, start, ok, _ = self._select(state, state.trial_vectors, trial_fitness)
        okf = ok.to(torch.float32)
        n_ok = okf.sum()
        if self.blocks > 1:
            # the success mask is shared by the blocks (one fitness) or one column per block, the parameters are per
            # block: sums over individuals
            Fm, CRm = state.F_vect.reshape(self.pop_size, -1), state.CR_vect.reshape(self.pop_size, -1)
            okm = okf.reshape(self.pop_size, -1)
            n_ok = okm.sum(0)
            sF, sF2 = (okm * Fm).sum(0), (okm * Fm ** 2).sum(0)
            mean_cr = (okm * CRm).sum(0) / n_ok
        else:
            sF, sF2 = (okf * state.F_vect).sum(), (okf * state.F_vect ** 2).sum()
            mean_cr = (okf * state.CR_vect).sum() / n_ok
        lehmer = sF2 / sF
        # with no success the reference's nansum/nanmean give NaN and poison the means; keep
        # the previous means instead
        F_u = torch.where(n_ok > 0, (1 - self.c) * state.F_u + self.c * _nan_to(lehmer), state.F_u)
        CR_u = torch.where(n_ok > 0, (1 - self.c) * state.CR_u + self.c * _nan_to(mean_cr), state.CR_u)
        return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start, start_index_dev=self._next_start_dev(state),
                             F_u=F_u, CR_u=CR_u)


class SHADE(JaDE):
    """SHADE (Tanabe & Fukunaga 2013): JADE's variation with a success-history memory of
    (F, CR) pairs updated by improvement-weighted means.  ``memory_size`` is the number of
    history slots H (the reference fixes 100 slots while sampling indices below ``pop_size``,
    ``shade.py:58,78,88``; here H is an explicit parameter)."""

    strict_improvement = True

    def __init__(self, lb, ub, pop_size=100, diff_padding_num=9, memory_size=100):
        self._init_common(lb, ub, pop_size)
        self.diff_padding_num = diff_padding_num
        self.H = int(memory_size)
        self.num_diff_vects = 1

    def setup(self, key):
        dev = self.lb.device
        shape = (2, self.H) if self.blocks == 1 else (2, self.blocks, self.H)      # a blocked instance: one memory per block
        return self._base_state(key, Memory_FCR=torch.full(shape, 0.5, device=dev),
                                F_vect=torch.zeros(self.pop_size * self.blocks, device=dev),
                                CR_vect=torch.zeros(self.pop_size * self.blocks, device=dev))

    def ask(self, state):
        key, k_var, k_slot, kF, kCR = rnd.split(state.key, 5)
        if self.blocks > 1:
            B = self.blocks
            slot = rnd.randint(k_slot, (self.pop_size * B,), 0, self.H)
            at = torch.arange(B, device=slot.device).repeat(self.pop_size) * self.H + slot      # (block, slot) of every virtual row
            F, CR = self._draw_params(state, kF, kCR, state.Memory_FCR[0].reshape(-1)[at], state.Memory_FCR[1].reshape(-1)[at])
        elif _adapt.supported(state.population):
            F, CR = _adapt.draw_params(k_slot, kF, kCR, state.Memory_FCR[0], state.Memory_FCR[1], self.pop_size)
        else:
            slot = rnd.randint(k_slot, (self.pop_size,), 0, self.H)
            F, CR = self._draw_params(state, kF, kCR, state.Memory_FCR[0, slot], state.Memory_FCR[1, slot])
        idx = self._window(state)
        trial = self._strategy_trials(k_var, state, idx, CURRENT, PBEST, self.num_diff_vects, BIN, F, CR,
                                      self.diff_padding_num)
        return trial, state.replace(trial_vectors=trial, key=key, F_vect=F, CR_vect=CR)

    def tell(self, state, trial_fitness):
        if self._fused_tell_ok(state):
            population, fitness, best_index, _, _, memory = _adapt.select_and_adapt(
                state.population, state.trial_vectors, state.fitness, trial_fitness, state.F_vect, state.CR_vect,
                state.Memory_FCR[0], state.Memory_FCR[1], strict=self.strict_improvement, weight_by_gain=True,
                mode="shade")
            return state.replace(population=population, fitness=fitness, best_index=best_index,
                                 Memory_FCR=memory)
        population, fitness, best_index, start, ok, parent_fit = self._select(state, state.trial_vectors, trial_fitness)
        if parent_fit.dim() == 2 and trial_fitness.dim() == 1:          # one fitness column per block, told one vector
            trial_fitness = trial_fitness[:, None].expand_as(parent_fit)
        delta = torch.where(ok, parent_fit - trial_fitness, torch.zeros_like(parent_fit))
        delta = torch.where(torch.isfinite(delta), delta, torch.zeros_like(delta))
        if self.blocks > 1:
            # improvements (hence weights) are shared by the blocks or one column per block; every block averages its
            # own parameters
            Fm, CRm = state.F_vect.reshape(self.pop_size, -1), state.CR_vect.reshape(self.pop_size, -1)
            dm = delta.reshape(self.pop_size, -1)
            total = dm.sum(0)
            w = dm / total
            M_CR = (w * CRm).sum(0)
            M_F = (w * Fm ** 2).sum(0) / (w * Fm).sum(0)
            rolled = torch.cat([torch.stack([M_F, M_CR])[:, :, None], state.Memory_FCR[:, :, :-1]], dim=2)
            valid = (total > 0) & torch.isfinite(M_F) & torch.isfinite(M_CR)
            memory = torch.where(valid[None, :, None], rolled, state.Memory_FCR)
            return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start,
                                 start_index_dev=self._next_start_dev(state), Memory_FCR=memory)
        total = delta.sum()
        w = delta / total
        M_CR = (w * state.CR_vect).sum()
        M_F = (w * state.F_vect ** 2).sum() / (w * state.F_vect).sum()
        rolled = torch.roll(state.Memory_FCR, 1, dims=1)
        rolled = torch.cat([torch.stack([M_F, M_CR])[:, None], rolled[:, 1:]], dim=1)
        valid = (total > 0) & torch.isfinite(M_F) & torch.isfinite(M_CR)
        memory = torch.where(valid, rolled, state.Memory_FCR)
        return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start, start_index_dev=self._next_start_dev(state),
                             Memory_FCR=memory)


def _newest_first(sel, values, depth):
    """Per column of the mask ``sel`` (rows, columns): the ``values`` of its selected rows, highest row index first, the
    first ``depth`` of them, NaN-padded — without a sort: a selected row's place is the number of selected rows behind it
    (a reversed cumulative sum), and the places of a column are distinct, so one scatter writes them (rows that are not
    selected, or lie deeper than ``depth``, go to a spare row)."""
    rows, cols = sel.shape
    st = sel.t().contiguous()                                                    # scans run along the contiguous dimension
    counts = st.to(torch.int32)
    behind = (counts.sum(1, keepdim=True) - torch.cumsum(counts, dim=1)).to(torch.int64)     # selected rows with a higher index
    place = torch.where(st & (behind < depth), behind, torch.full_like(behind, depth))
    buf = torch.full((cols, depth + 1), float("nan"), dtype=values.dtype, device=values.device)
    buf.scatter_(1, place, values[None, :].expand(cols, rows))
    return buf[:, :depth].t()


# strategy tables: (primary base, secondary base, #difference vectors, crossover)
_RAND_1_BIN = (RAND, RAND, 1, BIN)
_RAND2BEST_2_BIN = (RAND, BEST, 2, BIN)
_RAND_2_BIN = (RAND, RAND, 2, BIN)
_CURRENT2RAND_1 = (RAND, RAND, 1, ARITH)


class SaDE(DEBase):
    """SaDE (Qin, Huang & Suganthan 2009): four strategies chosen with probabilities learned
    from success/failure counts over the last ``LP`` generations; CR per strategy drawn around
    the median of its recent successful values; F ~ N(0.5, 0.3)."""

    strict_improvement = False

    def __init__(self, lb, ub, pop_size=100, diff_padding_num=9, LP=50):
        self._init_common(lb, ub, pop_size)
        self.diff_padding_num = diff_padding_num
        self.LP = LP
        pool = [_RAND_1_BIN, _RAND2BEST_2_BIN, _RAND_2_BIN, _CURRENT2RAND_1]
        self.strategy_pool = torch.tensor(pool, device=self.lb.device)
        self._code_sets = {"num_diff": {p[2] for p in pool}, "cross": {p[3] for p in pool}}

    def setup(self, key):
        dev = self.lb.device
        return self._base_state(
            key, success_memory=torch.zeros((self.LP, 4), device=dev), failure_memory=torch.zeros((self.LP, 4), device=dev),
            CR_memory=torch.full((self.LP, 4), float("nan"), device=dev), CRs=torch.zeros(self.batch_size, device=dev),
            strategy_ids=torch.zeros(self.batch_size, dtype=torch.int64, device=dev), iter=0)

    def graph_host_counters(self):
        return tuple(super().graph_host_counters()) + ("iter",)

    def graph_variant(self, state):
        # the learning-period branches of ask(): before, at and after the end of the first period
        return (state.iter >= self.LP, state.iter > self.LP)

    def ask(self, state):
        key, k_var, k_strat, k_cr, k_cr2, k_f = rnd.split(state.key, 6)
        dev = self.lb.device
        if state.iter >= self.LP:
            s, f = state.success_memory.sum(0), state.failure_memory.sum(0)
            rate = _nan_to(s / (s + f)) + 0.01
            p = rate / rate.sum()
        else:
            p = torch.full((4,), 0.25, device=dev)
        if state.iter > self.LP:
            crm = _nan_to(torch.nanmedian(state.CR_memory, dim=0).values, 0.5)
        else:
            crm = torch.full((4,), 0.5, device=dev)
        sid = rnd.choice(k_strat, 4, (self.batch_size,), p=p)
        cr_all = rnd.normal(k_cr, (self.batch_size, 4)) * 0.1 + crm
        cr_fix = rnd.normal(k_cr2, (self.batch_size, 4)) * 0.1 + crm
        cr_all = torch.where((cr_all < 0) | (cr_all > 1), cr_fix, cr_all)
        CR = cr_all[torch.arange(self.batch_size, device=dev), sid]
        F = rnd.normal(k_f, (self.batch_size,)) * 0.3 + 0.5
        code = self.strategy_pool[sid]
        idx = self._window(state)
        trial = self._strategy_trials(k_var, state, idx, code[:, 0], code[:, 1], code[:, 2], code[:, 3], F, CR,
                                      self.diff_padding_num, code_sets=self._code_sets)
        return trial, state.replace(trial_vectors=trial, key=key, CRs=CR, strategy_ids=sid, iter=state.iter + 1)

    def tell(self, state, trial_fitness):
        population, fitness, best_index, start, ok, _ = self._select(state, state.trial_vectors, trial_fitness)
        onehot = torch.nn.functional.one_hot(state.strategy_ids, 4).to(torch.float32)
        succ = (onehot * ok[:, None]).sum(0)
        fail = (onehot * (~ok)[:, None]).sum(0)
        success_memory = torch.cat([succ[None], state.success_memory[:-1]], dim=0)
        failure_memory = torch.cat([fail[None], state.failure_memory[:-1]], dim=0)
        # per strategy, push the successful CR values of this generation (most recent first) onto its
        # column of the LP-deep memory — all four columns at once
        B, LP, dev = state.CRs.shape[0], self.LP, state.CRs.device
        sel = ok[:, None] & onehot.bool()                                         # (B, 4)
        vals = _newest_first(sel, state.CRs, LP)                                  # (LP, 4), NaN where a column has fewer
        cnt = sel.sum(0).clamp(max=LP)                                            # (4,)
        pos = torch.arange(LP, device=dev)[:, None]
        shifted = torch.gather(state.CR_memory, 0, (pos - cnt).clamp_min(0))
        CR_memory = torch.where(pos < cnt, vals, shifted)
        return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start, start_index_dev=self._next_start_dev(state),
                             success_memory=success_memory, failure_memory=failure_memory, CR_memory=CR_memory)


class CoDE(DEBase):
    """CoDE (Wang, Cai & Zhang 2011): every individual gets three trial vectors (rand/1/bin,
    rand/2/bin, current-to-rand/1), each with control parameters drawn from a small pool; the
    best of the three competes with the parent."""

    strict_improvement = False

    def __init__(self, lb, ub, pop_size=100, batch_size=None, diff_padding_num=5, param_pool=None, replace=False):
        self._init_common(lb, ub, pop_size, batch_size)
        pool = [[1, 0.1], [1, 0.9], [0.8, 0.2]] if param_pool is None else param_pool
        self.param_pool = torch.as_tensor(pool, dtype=torch.float32).to(self.lb.device)
        self.diff_padding_num = diff_padding_num
        self.strategies = (_RAND_1_BIN, _RAND_2_BIN, _CURRENT2RAND_1)
        self.replace = replace

    def setup(self, key):
        return self._base_state(key, trial_rows=3 * self.batch_size)

    def ask(self, state):
        key, k_var, k_par = rnd.split(state.key, 3)
        idx = self._window(state)
        keys = rnd.split(k_var, 3)
        # one parameter pair per trial vector — of a blocked instance: per block of every trial vector
        pid = rnd.randint(k_par, (3, self.batch_size * self.blocks), 0, self.param_pool.shape[0])
        trials = []
        for s, (prim, sec, nd, cross) in enumerate(self.strategies):
            par = self.param_pool[pid[s]]
            trials.append(self._strategy_trials(keys[s], state, idx, prim, sec, nd, cross, par[:, 0], par[:, 1],
                                                self.diff_padding_num, self.replace))
        trial = torch.cat(trials, dim=0)
        return trial, state.replace(trial_vectors=trial, key=key)

    def tell(self, state, trial_fitness):
        b = self.batch_size
        fit3 = trial_fitness.reshape(3, b)
        which = torch.argmin(fit3, dim=0)
        cols = torch.arange(b, device=fit3.device)
        best_fit = fit3[which, cols]
        best_vec = state.trial_vectors.reshape(3, b, self.dim)[which, cols]
        population, fitness, best_index, start, _, _ = self._select(state, best_vec, best_fit)
        return state.replace(population=population, fitness=fitness, best_index=best_index, start_index=start, start_index_dev=self._next_start_dev(state))
