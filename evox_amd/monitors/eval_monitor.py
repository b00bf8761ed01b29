"""Evaluation monitor: running best / Pareto archive / fitness history.

The monitor never forces a device→host synchronisation inside the optimisation loop: the
running top-k is kept in device tensors and updated with a handful of small kernels, and the
per-generation fitness history keeps device tensors (or, with ``host_history=True``, copies
them to pinned host memory on a side stream).  Values are read back only when a getter is
called.  API parity: reference ``src/evox/monitors/eval_monitor.py:15-214``.
"""
from __future__ import annotations

import warnings

import torch

from ..core.monitor import Monitor


class EvalMonitor(Monitor):
    """
    full_fit_history   keep the fitness of every generation (default True).
    full_sol_history   keep every evaluated population (default False; costs memory).
    topk               single-objective: how many elite solutions to track.
    calc_pf            multi-objective: maintain an unbounded non-dominated archive.
    host_history       move history entries to pinned host memory asynchronously.
    """

    def __init__(self, full_fit_history=True, full_sol_history=False, topk=1, calc_pf=False, host_history=False):
        self.full_fit_history = full_fit_history
        self.full_sol_history = full_sol_history
        self.topk = topk
        self.calc_pf = calc_pf
        self.host_history = host_history
        self.fitness_history = []
        self.solution_history = []
        self.topk_fitness = None
        self.topk_solutions = None
        self.pf_solutions = None
        self.pf_fitness = None
        self.latest_solution = None
        self.latest_fitness = None
        self.eval_count = 0
        self.opt_direction = 1
        self._copy_stream = None
        self._pending = []

    def hooks(self):
        return ["post_eval"]

    def set_opt_direction(self, opt_direction):
        self.opt_direction = opt_direction

    # ------------------------------------------------------------------------------------
    def _archive(self, t: torch.Tensor):
        if not self.host_history or not t.is_cuda:
            return t
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=t.device)
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            host.copy_(t, non_blocking=True)
            t.record_stream(self._copy_stream)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._pending.append(done)
        return host

    def post_eval(self, _state, cand_sol, _transformed_cand_sol, fitness):
        if fitness.dim() == 1:
            self.record_fit_single_obj(cand_sol, fitness)
        else:
            self.record_fit_multi_obj(cand_sol, fitness)

    def record_fit_single_obj(self, cand_sol, fitness):
        self.eval_count += fitness.shape[0]
        self.latest_fitness, self.latest_solution = fitness, cand_sol
        if self.full_sol_history:
            self.solution_history.append(self._archive(cand_sol))
        if self.full_fit_history:
            self.fitness_history.append(self._archive(fitness))
        n_sol = cand_sol.shape[0] if isinstance(cand_sol, torch.Tensor) else None
        # in sharded evaluation a rank holds its slice of the candidates but the global
        # fitness: track the best among the rows it can see
        local_fit = fitness
        if n_sol is not None and n_sol != fitness.shape[0]:
            from ..core.distributed import get_process_id

            r = get_process_id()
            local_fit = fitness[r * n_sol:(r + 1) * n_sol]
        k = min(self.topk, local_fit.shape[0])
        if k == 1:
            fit, idx = torch.min(local_fit, dim=0, keepdim=True)
        else:
            fit, idx = torch.topk(local_fit, k, largest=False)
        sol = cand_sol[idx] if isinstance(cand_sol, torch.Tensor) else None
        if self.topk_fitness is None:
            self.topk_fitness, self.topk_solutions = fit, sol
            return
        if self.topk == 1:
            better = fit < self.topk_fitness
            self.topk_fitness = torch.where(better, fit, self.topk_fitness)
            if sol is not None:
                self.topk_solutions = torch.where(better[:, None], sol, self.topk_solutions)
        else:
            all_fit = torch.cat([self.topk_fitness, fit])
            keep = torch.topk(all_fit, min(self.topk, all_fit.shape[0]), largest=False).indices
            self.topk_fitness = all_fit[keep]
            if sol is not None:
                self.topk_solutions = torch.cat([self.topk_solutions, sol])[keep]

    def record_fit_multi_obj(self, cand_sol, fitness):
        self.eval_count += fitness.shape[0]
        if self.full_sol_history:
            self.solution_history.append(self._archive(cand_sol))
        if self.full_fit_history:
            self.fitness_history.append(self._archive(fitness))
        if self.calc_pf:
            from ..operators.selection.non_dominate import non_dominated_sort

            if self.pf_fitness is None:
                self.pf_fitness, self.pf_solutions = fitness, cand_sol
            else:
                self.pf_fitness = torch.cat([self.pf_fitness, fitness], dim=0)
                self.pf_solutions = torch.cat([self.pf_solutions, cand_sol], dim=0)
            front = non_dominated_sort(self.pf_fitness) == 0
            self.pf_fitness = self.pf_fitness[front]
            self.pf_solutions = self.pf_solutions[front]
        self.latest_fitness = fitness
        self.latest_solution = cand_sol

    # ------------------------------------------------------------------------------------
    def _dir(self, t):
        d = self.opt_direction
        if isinstance(d, torch.Tensor):
            return t * d.to(t.device)
        return t if d == 1 else t * d

    def get_latest_fitness(self):
        return self._dir(self.latest_fitness)

    def get_latest_solution(self):
        return self.latest_solution

    def get_pf_fitness(self):
        return self._dir(self.pf_fitness)

    def get_pf_solutions(self):
        return self.pf_solutions

    def get_topk_fitness(self):
        return self._dir(self.topk_fitness)

    def get_topk_solutions(self):
        return self.topk_solutions

    def get_best_solution(self):
        return self.topk_solutions[0]

    def get_best_fitness(self):
        return self._dir(self.topk_fitness)[0]

    def get_history(self):
        self.flush()
        return [self._dir(f) for f in self.fitness_history]

    def plot(self, problem_pf=None, **kwargs):
        if not self.fitness_history:
            warnings.warn("No fitness history recorded, return None")
            return None
        from ..vis_tools import plot

        hist = [h.detach().cpu().numpy() for h in self.get_history()]
        n_objs = 1 if hist[0].ndim == 1 else hist[0].shape[1]
        if n_objs == 1:
            return plot.plot_obj_space_1d(hist, **kwargs)
        if n_objs == 2:
            return plot.plot_obj_space_2d(hist, problem_pf, **kwargs)
        if n_objs == 3:
            return plot.plot_obj_space_3d(hist, problem_pf, **kwargs)
        warnings.warn("Not supported yet.")
        return None

    def flush(self):
        for ev in self._pending:
            ev.synchronize()
        self._pending.clear()

    def close(self):
        self.flush()
