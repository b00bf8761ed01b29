"""The two legacy monitors kept for API compatibility (the reference deprecates both in favour
of ``EvalMonitor``): ``src/evox/monitors/std_so_monitor.py:8-126`` and
``std_mo_monitor.py:10-110``."""
from __future__ import annotations

import warnings

import torch

from ..core.monitor import Monitor
from ..operators.selection.non_dominate import non_dominated_sort


class _LegacyMonitor(Monitor):
    def __init__(self, record_fit_history, record_pop_history):
        self.record_fit_history = record_fit_history
        self.record_pop_history = record_pop_history
        self.fitness_history = []
        self.population_history = []
        self.current_population = None
        self.opt_direction = 1

    def set_opt_direction(self, opt_direction):
        self.opt_direction = opt_direction

    def hooks(self):
        return ["post_ask", "post_eval"]

    def post_ask(self, _state, cand_sol):
        self.record_pop(cand_sol)

    def post_eval(self, _state, _cand_sol, _transformed_cand_sol, fitness):
        self.record_fit(fitness)

    def record_pop(self, pop, tranform=None):
        if self.record_pop_history:
            self.population_history.append(pop)
        self.current_population = pop

    def get_last(self):
        return self.opt_direction * self.fitness_history[-1]

    def get_history(self):
        return [self.opt_direction * fit for fit in self.fitness_history]

    def flush(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def close(self):
        self.flush()


class StdSOMonitor(_LegacyMonitor):
    """Tracks the ``record_topk`` best individuals seen so far (on-device merge + sort)."""

    def __init__(self, record_topk=1, record_fit_history=True, record_pop_history=False):
        warnings.warn("The StdSOMonitor is deprecated in favor of the new EvalMonitor.", DeprecationWarning)
        super().__init__(record_fit_history, record_pop_history)
        self.record_topk = record_topk
        self.topk_solutions = None
        self.topk_fitness = None

    def record_fit(self, fitness, metrics=None, transform=None):
        if self.record_fit_history:
            self.fitness_history.append(fitness)
        have_pop = self.current_population is not None
        fit = fitness if self.topk_fitness is None else torch.cat([self.topk_fitness, fitness])
        if have_pop:
            sol = self.current_population if self.topk_solutions is None else torch.cat(
                [self.topk_solutions, self.current_population], dim=0)
        order = torch.argsort(fit, stable=True)[: self.record_topk]
        self.topk_fitness = fit[order]
        if have_pop:
            self.topk_solutions = sol[order]

    def get_topk_fitness(self):
        return self.opt_direction * self.topk_fitness

    def get_topk_solutions(self):
        return self.topk_solutions

    def get_best_fitness(self):
        if self.topk_fitness is None:
            warnings.warn("trying to get info from a monitor with no recorded data")
            return None
        return self.opt_direction * self.topk_fitness[0]

    def get_best_solution(self):
        if self.topk_solutions is None:
            warnings.warn("trying to get info from a monitor with no recorded data")
            return None
        return self.topk_solutions[0]


class StdMOMonitor(_LegacyMonitor):
    """Optionally maintains the unbounded non-dominated archive of everything evaluated."""

    def __init__(self, record_pf=False, record_fit_history=True, record_pop_history=False):
        warnings.warn("The StdMOMonitor is deprecated in favor of the new EvalMonitor.", DeprecationWarning)
        super().__init__(record_fit_history, record_pop_history)
        self.record_pf = record_pf
        self.pf_solutions = None
        self.pf_fitness = None

    def record_fit(self, fitness, metrics=None, tranform=None):
        if self.record_fit_history:
            self.fitness_history.append(fitness)
        if not self.record_pf:
            return
        fit = fitness if self.pf_fitness is None else torch.cat([self.pf_fitness, fitness], dim=0)
        have_pop = self.current_population is not None
        if have_pop:
            sol = self.current_population if self.pf_solutions is None else torch.cat(
                [self.pf_solutions, self.current_population], dim=0)
        front = non_dominated_sort(fit.to(torch.float32)) == 0
        self.pf_fitness = fit[front]
        if have_pop:
            self.pf_solutions = sol[front]

    def get_pf_fitness(self):
        return self.opt_direction * self.pf_fitness

    def get_pf_solutions(self):
        return self.pf_solutions
